# Where the ping-pong kernel's K step goes: timing builds of gemm_f16s.hip (results wrong by construction) with one ingredient removed,
# on the body's GEMM shapes, cold (40 rotating weight copies) and hot (one copy) weights.  `tools/pp_ablation.sh build` makes the
# libraries in tools/build/ (they travel to the GPU box); output gpurun_out/r05/pp_ablation.log
set -e
if [ "$1" = build ]; then
  cd "$(dirname "$0")/../artalk_amd/csrc" && mkdir -p ../../tools/build/abl
  for v in NOMFMA NODMA NOLDS; do
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -I../../include -DPP_ABL_$v -c gemm_f16s.hip -o ../../tools/build/abl/gemm_pp_$v.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/build/libartalk_pp_$v.so build/gemm_f32.o ../../tools/build/abl/gemm_pp_$v.o build/attention.o build/norm.o \
        build/w2v_front.o build/ar_glue.o build/flame.o build/engine.o -L/opt/rocm/lib -lrocprofiler-sdk-roctx -Wl,-rpath,/opt/rocm/lib ) &
  done
  wait; exit 0
fi
O=gpurun_out/r05/pp_ablation.log; mkdir -p gpurun_out/r05; : > $O
export GEMM_ONLY="t1600 qkv,t1600 ffn1,t3200 qkv,t3200 ffn1" GEMM_VARIANTS="28:1,31:1,30:1,33:1"
for rot in 40 1; do
  export GEMM_ROTATE=$rot
  for v in base NOMFMA NODMA NOLDS; do
    if [ $v = base ]; then unset ARTALK_LIB; else export ARTALK_LIB=$PWD/tools/build/libartalk_pp_$v.so; fi
    echo "== $v (rotate $rot)" >> $O
    GEMM_ROUNDS=5 timeout -k 10 120 python3 tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids >> $O
  done
done
cat $O
