# Tile configuration sweep for the AR-block GEMMs at the row counts of the large scale steps (cold weights, graph replay).
export GEMM_ROTATE=40
O=gpurun_out/r02_mid_gemm_sweep.log
: > $O
for M in 400 800 1600 3200; do
  GEMM_ONLY="t$M qkv,t$M proj,t$M ffn1" GEMM_VARIANTS="276:1,277:1,278:1,264:1,532:1,520:1,281:1,282:1" timeout -k 10 300 python tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids >> $O
  GEMM_ONLY="t$M ffn2" GEMM_VARIANTS="276:1,788:1,1556:1,277:1,278:1,264:1,520:1,776:1,1032:1,793:1" timeout -k 10 300 python tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
