cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_levels
rm -rf $O && mkdir -p $O
for B in 32 16; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_b$B -- python3 bench.py --steps 2 --warmup 2 --batch $B --branches 1 --resident --no-cpu-baseline --no-alt-mode > $O/kt_b$B.log 2>&1
python tools/summarize_profile.py levels $O/kt_b$B $O/levels_b$B.csv
find $O/kt_b$B -name "*kernel_trace.csv" -delete
done
