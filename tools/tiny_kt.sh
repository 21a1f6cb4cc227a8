cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_tiny_kt
rm -rf $O && mkdir -p $O
GEMM_GRAPH=0 GEMM_ROTATE=40 GEMM_ONLY="t80 qkv,t80 proj,t160 ffn1" GEMM_VARIANTS="276:1,791:1,1560:1,1048:1" timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 tools/gemm_f16s_bench.py > $O/log.txt 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r02_tiny_kt/kt/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"].replace("void ","").replace("artalk::","").split("(")[0][:60], int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), r.get("LDS_Block_Size",""), r.get("VGPR_Count",""), r.get("SGPR_Count",""))
    agg[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k, v in sorted(agg.items()):
    v.sort()
    print(k, "n=%d min=%.2f med=%.2f max=%.2f" % (len(v), v[0], v[len(v)//2], v[-1]))
PY
find $O -name "*kernel_trace.csv" -delete
