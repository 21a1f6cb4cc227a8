# Tile configuration sweep for the AR-block GEMMs at the row counts of the large scale steps and the VAE stacks (cold weights, graph
# replay): small-grid 64x64 kernel (20) against the mid-grid 128x128 kernel (28), unsplit and split over K.  Output: gpurun_out/r03/mid_gemm_sweep.log
export GEMM_ROTATE=40
O=gpurun_out/r03/mid_gemm_sweep.log
mkdir -p gpurun_out/r03
: > $O
for M in 800 1600 3200; do
  GEMM_ONLY="t$M qkv,t$M ffn1" GEMM_VARIANTS="20:1,28:1" timeout -k 10 300 python tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids >> $O
  GEMM_ONLY="t$M proj" GEMM_VARIANTS="20:1,28:1,540:1,796:1" timeout -k 10 300 python tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids >> $O
  GEMM_ONLY="t$M ffn2" GEMM_VARIANTS="20:1,788:1,28:1,540:1,796:1,1052:1" timeout -k 10 300 python tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
