# Where the encoder attention kernel's time goes (attention_f16_wide_kernel, 199 x 199 per head, 1536 heads per launch): timing builds of
# attention.hip (results wrong by construction) with one ingredient removed each: ATT_ABL 1 no K / V loads from global memory, 2 no softmax
# arithmetic (max / exp / rescale), 3 one MFMA product of three, 4 no hi / lo split of P.  `tools/attn_ablation.sh build` makes them in
# tools/build/ (travels to the GPU box); without arguments it times the model's launch shape with every build.
set -e
if [ "$1" = build ]; then
  cd "$(dirname "$0")/../artalk_amd/csrc" && mkdir -p ../../tools/build/abl
  for v in 1 2 3 4; do
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -I../../include -DATT_ABL=$v -c attention.hip -o ../../tools/build/abl/attention_$v.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/build/libartalk_att$v.so build/gemm_f32.o build/gemm_f16s.o ../../tools/build/abl/attention_$v.o build/norm.o \
        build/w2v_front.o build/ar_glue.o build/flame.o build/engine.o -L/opt/rocm/lib -lrocprofiler-sdk-roctx -Wl,-rpath,/opt/rocm/lib ) &
  done
  wait; exit 0
fi
for v in base 1 2 3 4 base; do
  if [ $v = base ]; then unset ARTALK_LIB; else export ARTALK_LIB=$PWD/tools/build/libartalk_att$v.so; fi
  echo "== ATT_ABL $v"
  ATTN_ONLY=w2v timeout -k 10 120 python3 tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
done
