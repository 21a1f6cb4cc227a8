# Collects everything kept under profiles/ for a round: run on the GPU box as
#   gpurun -- 'bash tools/collect_profiles.sh r02'   (outputs under gpurun_out/<round>/, then copied into profiles/ by hand)
set -e
R=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
rm -rf $O && mkdir -p $O
echo "== kernel trace f16x3"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_f16x3 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-mode > $O/kt_f16x3.log 2>&1
python tools/summarize_profile.py stats $O/kt_f16x3 $O/f16x3_kernel_stats.csv > /dev/null
python tools/summarize_profile.py shapes $O/kt_f16x3 $O/f16x3_launch_shapes.csv > /dev/null
find $O/kt_f16x3 -name "*kernel_trace.csv" -delete
echo "== kernel trace, one clip group: the AR/VAE body per scale step"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_b32 -- python3 bench.py --steps 2 --warmup 2 --branches 1 --no-cpu-baseline --no-alt-mode > $O/kt_b32.log 2>&1
python tools/summarize_profile.py levels $O/kt_b32 $O/body_levels_b32.csv > /dev/null
find $O/kt_b32 -name "*kernel_trace.csv" -delete
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_b16 -- python3 bench.py --steps 2 --warmup 2 --batch 16 --branches 1 --no-cpu-baseline --no-alt-mode > $O/kt_b16.log 2>&1
python tools/summarize_profile.py levels $O/kt_b16 $O/body_levels_b16.csv > /dev/null
find $O/kt_b16 -name "*kernel_trace.csv" -delete
echo "== roctx ranges (marker trace)"
timeout -k 10 300 rocprofv3 --marker-trace --output-format csv -d $O/mk -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-mode > $O/mk.log 2>&1 || true
python tools/summarize_profile.py markers $O/mk $O/roctx_ranges.csv > /dev/null || true
find $O/mk -name "*marker_api_trace.csv" -delete || true
echo "== kernel trace f32"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_f32 -- python3 bench.py --steps 3 --warmup 2 --precision f32 --no-cpu-baseline --no-alt-mode > $O/kt_f32.log 2>&1
python tools/summarize_profile.py stats $O/kt_f32 $O/f32_kernel_stats.csv > /dev/null
find $O/kt_f32 -name "*kernel_trace.csv" -delete
echo "== pmc fetch"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-mode > $O/pmc_fetch.log 2>&1
echo "== pmc write"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-mode > $O/pmc_write.log 2>&1
python tools/summarize_profile.py pmc $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json | head -12
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
echo "== unprofiled default bench (after the PMC passes: its roofline.traffic is read from THIS run's pmc_traffic.json)"
cp $O/pmc_traffic.json profiles/${R}_pmc_traffic.json
timeout -k 10 500 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/bench_line.json
cut -c1-400 $O/bench_line.json
echo "== gemm microbenchmarks"
GEMM_GRAPH=0 GEMM_VARIANTS="13:1,7:1,12:1,8:1,99:1" GEMM_ONLY="w2v qkv,w2v out,w2v ff1,w2v ff2,conv1,ada" timeout -k 10 300 python tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids > $O/gemm_f16s_bench.log
echo "# the same with the epilogues the model uses: P8 result + GELU (q|k|v: P8 only in the model), in-place residual" >> $O/gemm_f16s_bench.log
GEMM_ACT=0x101 GEMM_GRAPH=0 GEMM_VARIANTS="13:1,7:1,12:1,8:1,99:1" GEMM_ONLY="w2v qkv,w2v ff1" timeout -k 10 300 python tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids >> $O/gemm_f16s_bench.log
GEMM_ACT=0x200 GEMM_GRAPH=0 GEMM_VARIANTS="7:1,12:1,8:1,99:1" GEMM_ONLY="w2v out,w2v ff2" timeout -k 10 300 python tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids >> $O/gemm_f16s_bench.log
timeout -k 10 100 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids > $O/attn_bench.log
echo "# the encoder shape without the persistent ping-pong kernel (ARTALK_ATTN_PP=0: one head per workgroup, keys staged once)" >> $O/attn_bench.log
ARTALK_ATTN_PP=0 ATTN_ONLY=w2v timeout -k 10 100 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids >> $O/attn_bench.log
echo "== matrix-pipe duty per instantiation of the dominant kernel (PMC passes of the real step)"
bash tools/pmc_mfma_bench.sh $R > /dev/null 2>&1 || echo "mfma pmc failed"
cat $O/mfma_util_per_instantiation.log
echo "== P8 headroom"
timeout -k 10 200 python tools/p8_headroom.py $O/p8_headroom.json 2>&1 | grep -v amdgpu.ids | tail -14
ls $O
echo "== P8 headroom, heavy profile, after the calibration pass (per-site scales)"
timeout -k 10 300 python tools/p8_headroom.py $O/p8_headroom_heavy.json --profile heavy --calibrate 2>&1 | grep -v amdgpu.ids | tail -14
ls $O
