# Collects everything kept under profiles/ for a round: run on the GPU box as
#   gpurun -- 'bash tools/collect_profiles.sh'   (outputs under gpurun_out/r01/, then copied into profiles/ by hand)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01
rm -rf $O && mkdir -p $O
echo "== unprofiled default bench"
timeout -k 10 500 python bench.py 2>$O/bench.err | tail -1 > $O/bench_line.json
cut -c1-400 $O/bench_line.json
echo "== kernel trace f16x3"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_f16x3 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-mode > $O/kt_f16x3.log 2>&1
python tools/summarize_profile.py stats $O/kt_f16x3 $O/f16x3_kernel_stats.csv > /dev/null
python tools/summarize_profile.py shapes $O/kt_f16x3 $O/f16x3_launch_shapes.csv > /dev/null
find $O/kt_f16x3 -name "*kernel_trace.csv" -delete
echo "== kernel trace f32"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_f32 -- python3 bench.py --steps 5 --warmup 2 --precision f32 --no-cpu-baseline > $O/kt_f32.log 2>&1
python tools/summarize_profile.py stats $O/kt_f32 $O/f32_kernel_stats.csv > /dev/null
find $O/kt_f32 -name "*kernel_trace.csv" -delete
echo "== pmc fetch"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-mode > $O/pmc_fetch.log 2>&1
echo "== pmc write"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-mode > $O/pmc_write.log 2>&1
python tools/summarize_profile.py pmc $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json | head -12
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
echo "== gemm microbenchmarks"
GEMM_VARIANTS="1:1,2:1,7:1,8:1" GEMM_ONLY="w2v qkv,w2v out,w2v ff1,w2v ff2,conv1,ada" timeout -k 10 200 python tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids > $O/gemm_f16s_bench.log
GEMM_VARIANTS="1:1,20:1,21:1" GEMM_ONLY="g qkv p4,g proj p4,g ffn1 p4,g ffn2 p4,g qkv p2,g proj p2,g ffn1 p2,g ffn2 p2,g qkv p1,g ffn2 p1,g hist kv" timeout -k 10 200 python tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids >> $O/gemm_f16s_bench.log
for c in 16 17 18; do echo "-- STAMP_CFG=$c" >> $O/gemm_p8_stamps.log; STAMP_CFG=$c timeout -k 10 100 python tools/gemm_p8_stamps.py 2>&1 | grep -v amdgpu.ids >> $O/gemm_p8_stamps.log; done
timeout -k 10 100 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids > $O/attn_bench.log
timeout -k 10 60 python tools/mfma_subnormal_probe.py > $O/mfma_subnormal_probe.log 2>&1
ls $O
