cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for GM in 4 8 16; do
ARTALK_P8_GM=$GM python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-mode 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('GM=$GM', r['ms_per_step'], r['stages_ms']['w2v_encoder_ms'], r['roofline']['achieved'])"
O=gpurun_out/gm$GM; rm -rf $O
ARTALK_P8_GM=$GM timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-mode > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/**/*counter_collection.csv", recursive=True)[0]
n = 0; tot = 0.0
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and "gemm_p8_2wgp_kernel<0>" in r["Kernel_Name"]:
        n += 1; tot += float(r["Counter_Value"])
print("GM=$GM fetch MB/launch", 2 * tot / n * 1024 / 1e6, n)
PY
rm -rf $O
done
