# Sweep (ring depth, tile, split-K) of the small-grid split GEMM on the row counts of the AR scale steps, weights rotating through
# 40 copies (cold, as in the model).  Output: gpurun_out/r03_tiny_gemm_sweep.log
export GEMM_ROTATE=40
O=gpurun_out/r03_tiny_gemm_sweep.log
: > $O
for M in 16 32 80 160 400 800; do
  GEMM_ONLY="t$M qkv,t$M proj,t$M ffn1" GEMM_VARIANTS="20:1,532:1,1047:1,791:1,1560:1,1048:1,1559:1" timeout -k 10 300 python tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids >> $O
  GEMM_ONLY="t$M ffn2" GEMM_VARIANTS="1556:1,788:1,3095:1,4119:1,4120:1,2071:1,3092:1" timeout -k 10 300 python tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
