# MFMA utilisation of the big-tile GEMM (and the whole step's kernels) from hardware counters, in their own rocprofv3 passes
# (--pmc with --kernel-trace only).  SQ_VALU_MFMA_BUSY_CYCLES counts cycles a SIMD's matrix pipe is busy (= 32 x MFMAs issued for
# 32x32x16 f16), SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE give the denominator.  Output: gpurun_out/r03/pmc_mfma_*.csv
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
for C in "SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM" ; do
  GEMM_ROUNDS=1 GEMM_GRAPH=0 GEMM_VARIANTS="12:1" GEMM_ONLY="w2v qkv,w2v ff2" timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_mfma_$C -- python3 tools/gemm_f16s_bench.py > $O/pmc_mfma_$C.log 2>&1 || echo "counter $C failed"
done
python3 - <<'PY'
import csv, glob, collections, os
O="gpurun_out/r03"
res=collections.defaultdict(dict)
for d in glob.glob(O+"/pmc_mfma_*/"):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_p8_big" not in r["Kernel_Name"]: continue
            k=(r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X","?"))
            res[r["Counter_Name"]].setdefault(k,[]).append(float(r["Counter_Value"]))
for c,v in res.items():
    for k,vals in v.items():
        print(c, "grid", k, "n", len(vals), "median", sorted(vals)[len(vals)//2])
PY
