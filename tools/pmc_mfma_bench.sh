# Matrix-pipe duty and shader clock of EVERY instantiation of the dominant kernel (and of the other matrix kernels of the step) inside the real step (bench.py, one timed step), from
# hardware counters, each in its own rocprofv3 pass (--pmc with --kernel-trace only):
#   clock = GRBM_GUI_ACTIVE / 8 / duration        busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8)
# usage (GPU box): bash tools/pmc_mfma_bench.sh r04      -> gpurun_out/r04/mfma_util_per_instantiation.log
set -e
R=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
for C in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmcb_$C -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-mode > $O/pmcb_$C.log 2>&1 || echo "counter $C failed"
done
python3 - $O <<'PY' | tee $O/mfma_util_per_instantiation.log
import csv, glob, collections, sys
O=sys.argv[1]
KERNELS=("gemm_p8_big", "gemm_p8_mid", "gemm_p8_pp", "gemm_p8_sm_kernel<64, 64, 4", "gemm_p8_2wgp", "attention_f16_pp", "attention_f16_wide_ar", "posconv_p8")      # grids are reported in units of 512 threads
val=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
def key(r, gx): return (r["Kernel_Name"].split("(")[0].replace("void artalk::",""), int(gx))
for c in ("SQ_VALU_MFMA_BUSY_CYCLES","GRBM_GUI_ACTIVE"):
    for f in glob.glob(f"{O}/pmcb_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if any(k in r["Kernel_Name"] for k in KERNELS) and r["Counter_Name"]==c:
                val[c][key(r, int(r["Grid_Size"])//512)].append(float(r["Counter_Value"]))
    for f in glob.glob(f"{O}/pmcb_{c}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if any(k in r["Kernel_Name"] for k in KERNELS):
                dur[key(r, int(r["Grid_Size_X"])//512)].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
med=lambda a: sorted(a)[len(a)//2]
print("instantiation (workgroups)                         launches  median_us  shader_clock_GHz  mfma_busy_of_all_SIMDs")
for k in sorted(val["GRBM_GUI_ACTIVE"]):
    if k not in val["SQ_VALU_MFMA_BUSY_CYCLES"] or k not in dur: continue
    cyc=med(val["GRBM_GUI_ACTIVE"][k])/8; d=med(dur[k]); busy=med(val["SQ_VALU_MFMA_BUSY_CYCLES"][k])/1024/cyc
    print(f"{k[0]:42s} ({k[1]:3d})  {len(dur[k])//2:8d}  {d:9.1f}  {cyc/d/1e3:16.2f}  {busy:22.3f}")
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
