set -e
for cfgenv in "0 0" "20 0" "40 0" "80 0" "0 1" "40 1"; do
  set -- $cfgenv
  echo "== stagger_us=$1 dma_split=$2"
  ARTALK_P8_STAGGER_US=$1 ARTALK_P8_DMA_SPLIT=$2 STAMP_CFG=17 timeout -k 10 100 python tools/gemm_p8_stamps.py 2>&1 | grep -v amdgpu.ids | cut -c1-230
  ARTALK_P8_STAGGER_US=$1 ARTALK_P8_DMA_SPLIT=$2 GEMM_ONLY="w2v qkv,w2v ff1,conv1" GEMM_VARIANTS="7:1" timeout -k 10 100 python tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids
done
