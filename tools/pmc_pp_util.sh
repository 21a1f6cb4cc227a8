# Matrix-pipe duty and shader clock of the body's GEMM kernels (mid-grid 128x128: cfg 28; ping-pong 256x128: cfg 31; 128x128: cfg 33) from hardware
# counters, each counter in its own rocprofv3 pass; production build and the NODMA timing build (tools/pp_ablation.sh build).
#   clock = GRBM_GUI_ACTIVE / 8 / duration        busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 / (GRBM_GUI_ACTIVE / 8)
# Output: gpurun_out/r05/pp_mfma_util.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export GEMM_ROTATE=8 GEMM_ROUNDS=3 GEMM_GRAPH=0 GEMM_VARIANTS="28:1,31:1,33:1" GEMM_ONLY="t1600 qkv,t1600 ffn1,t3200 qkv,t3200 ffn1"
for v in base NODMA; do
  if [ $v = base ]; then unset ARTALK_LIB; else export ARTALK_LIB=$PWD/tools/build/libartalk_pp_$v.so; fi
  for C in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
    rm -rf $O/pmc_pp_${v}_$C
    timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_pp_${v}_$C -- python3 tools/gemm_f16s_bench.py > $O/pmc_pp_${v}_$C.log 2>&1 || echo "counter $C failed"
  done
done
python3 - <<'PY' | tee gpurun_out/r05/pp_mfma_util.log
import csv, glob, collections, re
O="gpurun_out/r05"
med=lambda a: sorted(a)[len(a)//2]
print("build  kernel                     workgroups  duration_us  shader_clock_GHz  mfma_busy_all_SIMDs")
for v in ("base","NODMA"):
    val=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES","GRBM_GUI_ACTIVE"):
        for f in glob.glob(f"{O}/pmc_pp_{v}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k=r["Kernel_Name"]
                if ("gemm_p8_pp" in k or "gemm_p8_mid" in k) and r["Counter_Name"]==c:
                    name=re.sub(r"artalk::|\(.*","",k)[:26]
                    val[c][(name,int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
        for f in glob.glob(f"{O}/pmc_pp_{v}_{c}/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k=r["Kernel_Name"]
                if "gemm_p8_pp" in k or "gemm_p8_mid" in k:
                    name=re.sub(r"artalk::|\(.*","",k)[:26]
                    dur[(name,int(r["Grid_Size_X"])*int(r.get("Grid_Size_Y",1) or 1))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    for key in sorted(val["GRBM_GUI_ACTIVE"]):
        if key not in dur or key not in val["SQ_VALU_MFMA_BUSY_CYCLES"]: continue
        cyc=med(val["GRBM_GUI_ACTIVE"][key])/8; d=med(dur[key]); busy=med(val["SQ_VALU_MFMA_BUSY_CYCLES"][key])/1024/cyc
        print(f"{v:6s} {key[0]:26s} {key[1]//512:10d}  {d:11.1f}  {cyc/d/1e3:16.2f}  {busy:19.3f}")
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
