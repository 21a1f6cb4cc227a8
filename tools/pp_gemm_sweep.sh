# Ping-pong kernel (cfg 30-35: gemm_p8_pp_kernel) against the small-grid (20) and mid-grid (28) kernels on the GEMM shapes of the large AR
# scale steps and the VAE stacks, per clip group of 16 and for a merged group of 32 (cold weights: 40 rotating copies, graph replay).
# Output: gpurun_out/r05/pp_gemm_sweep.log
export GEMM_ROTATE=40
O=gpurun_out/r05/pp_gemm_sweep.log
mkdir -p gpurun_out/r05
: > $O
for M in 800 1600 3200; do
  GEMM_ONLY="t$M qkv,t$M ffn1" GEMM_VARIANTS="20:1,28:1,30:1,31:1,32:1,33:1,34:1" timeout -k 10 300 python tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids >> $O
  GEMM_ONLY="t$M proj" GEMM_VARIANTS="20:1,28:1,30:1,32:1,542:1,798:1,544:1,800:1" timeout -k 10 300 python tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids >> $O
  GEMM_ONLY="t$M ffn2" GEMM_VARIANTS="788:1,796:1,30:1,542:1,798:1,1054:1,800:1,1056:1" timeout -k 10 300 python tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids >> $O
done
GEMM_ONLY="v qkv,v mlp1,w qkv,w mlp1,e qkv,e mlp1" GEMM_VARIANTS="20:1,28:1,30:1,31:1,32:1,34:1" timeout -k 10 300 python tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids >> $O
GEMM_ONLY="v proj,v mlp2,w proj,w mlp2,e proj,e mlp2" GEMM_VARIANTS="20:1,28:1,30:1,32:1,542:1,544:1" timeout -k 10 300 python tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids >> $O
cat $O
