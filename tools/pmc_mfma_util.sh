# MFMA utilisation and shader clock of the big-tile GEMM from hardware counters, each counter in its own rocprofv3 pass (--pmc with
# --kernel-trace only).  SQ_VALU_MFMA_BUSY_CYCLES sums, over the chip's 1024 SIMDs, the cycles a matrix pipe is busy (= 32 x MFMAs issued
# for v_mfma_f32_32x32x16_f16); GRBM_GUI_ACTIVE sums the busy shader-clock cycles of the 8 XCDs; the kernel trace gives the duration.
#   clock = GRBM_GUI_ACTIVE / 8 / duration        busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 / (GRBM_GUI_ACTIVE / 8)
# Shapes: the encoder's FFN-out GEMM at batch 32 (240 tiles of 320x256 on 256 CUs) and the same work per CU on 30 / 60 / 120 CUs.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
for C in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  GEMM_ROUNDS=3 GEMM_GRAPH=0 GEMM_VARIANTS="12:1" GEMM_ONLY="w2v qkv,p*" timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_mfma_$C -- python3 tools/gemm_f16s_bench.py > $O/pmc_mfma_$C.log 2>&1 || echo "counter $C failed"
done
python3 - <<'PY' | tee gpurun_out/r03/mfma_util.log
import csv, glob, collections
O="gpurun_out/r03"
val=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
for c in ("SQ_VALU_MFMA_BUSY_CYCLES","GRBM_GUI_ACTIVE"):
    for f in glob.glob(f"{O}/pmc_mfma_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_p8_big" in r["Kernel_Name"] and r["Counter_Name"]==c:
                val[c][int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    for f in glob.glob(f"{O}/pmc_mfma_{c}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_p8_big" in r["Kernel_Name"]:
                dur[int(r["Grid_Size_X"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
med=lambda a: sorted(a)[len(a)//2]
print("workgroups  duration_us  shader_clock_GHz  mfma_busy_of_all_SIMDs  mfma_busy_of_active_CUs")
for g in sorted(val["GRBM_GUI_ACTIVE"]):
    wg=g//512; cyc=med(val["GRBM_GUI_ACTIVE"][g])/8; d=med(dur[g]); busy=med(val["SQ_VALU_MFMA_BUSY_CYCLES"][g])/1024/cyc
    print(f"{wg:10d}  {d:11.1f}  {cyc/d/1e3:16.2f}  {busy:22.3f}  {busy*256/min(wg,256):23.3f}")
PY
