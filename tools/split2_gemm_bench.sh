# The two-launch answer to round quantisation per shape (appended to the round's gemm_f16s_bench.log): cfg12 = ONE launch of 320 x 256 tiles,
# cfg99 = the dispatch's own choice, which with ARTALK_P8_SPLIT2=1 is whole rounds of 320 x 256 + the remaining rows as 256 x 256 tiles where
# such a split exists (FFN-in: 768 + 240 tiles; q|k|v has none and stays one launch).  Model epilogues (P8 result + GELU).
O=gpurun_out/r05/split2_gemm_bench.log; mkdir -p gpurun_out/r05; : > $O
for sp in 0 1 0 1; do
  echo "# ARTALK_P8_SPLIT2=$sp" >> $O
  ARTALK_P8_SPLIT2=$sp GEMM_ACT=0x101 GEMM_GRAPH=0 GEMM_VARIANTS="12:1,99:1" GEMM_ONLY="w2v qkv,w2v ff1" timeout -k 10 200 python tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
