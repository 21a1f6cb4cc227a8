# Where the big-tile GEMM's time goes: timing builds of gemm_f16s.hip (results wrong, by construction) with one ingredient removed each,
# timed on the encoder's four GEMM shapes and the AdaLN table, plus the shader clock each build holds (GRBM_GUI_ACTIVE / 8 / duration,
# own --pmc pass).  Builds (made here by `tools/gemm_big_ablation.sh build`, in tools/build/, which travels to the GPU box):
#   NOLDS   no fragment reads from LDS (registers loaded once)      NODMA    no LDS-DMA pieces after a tile's first two stages
#   ONEPROD one MFMA product of three                               NOSTORE  epilogue stores masked off
#   phase16 / phasex  the first tile of a workgroup shortened by a per-CU / per-XCD phase, so that epilogues stop coinciding
set -e
if [ "$1" = build ]; then
  cd "$(dirname "$0")/../artalk_amd/csrc" && mkdir -p ../../tools/build/abl
  for v in NOLDS NODMA ONEPROD NOSTORE phase16 phasex mid_NOMFMA mid_NODMA mid_NOBAR mid_NOB; do
    case $v in phase16) D="-DBIG_PHASE=16";; phasex) D="-DBIG_PHASE=8 -DBIG_PHASE_XCD";; mid_*) D="-DMID_ABL_${v#mid_}";; *) D="-DBIG_ABL_$v";; esac
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -I../../include $D -c gemm_f16s.hip -o ../../tools/build/abl/gemm_$v.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/build/libartalk_$v.so build/gemm_f32.o ../../tools/build/abl/gemm_$v.o build/attention.o build/norm.o \
        build/w2v_front.o build/ar_glue.o build/flame.o build/engine.o -L/opt/rocm/lib -lrocprofiler-sdk-roctx -Wl,-rpath,/opt/rocm/lib ) &
  done
  wait; exit 0
fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/abl; mkdir -p $O
export GEMM_ONLY="w2v qkv,w2v out,w2v ff1,w2v ff2,ada" GEMM_VARIANTS="12:1"
for v in base NOLDS NODMA ONEPROD NOSTORE phase16 phasex base; do
  if [ $v = base ]; then unset ARTALK_LIB; else export ARTALK_LIB=$PWD/tools/build/libartalk_$v.so; fi
  echo "== $v"
  GEMM_ROUNDS=5 timeout -k 10 120 python3 tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids
  GEMM_ROUNDS=3 GEMM_GRAPH=0 timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/$v -- python3 tools/gemm_f16s_bench.py > $O/$v.log 2>&1 || echo "pmc pass failed"
  python3 - $O/$v <<'PY'
import csv, glob, sys, collections
val=collections.defaultdict(list); dur=collections.defaultdict(list); disp={}
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_p8_big" in r["Kernel_Name"] and r["Counter_Name"]=="GRBM_GUI_ACTIVE":
            val[r["Dispatch_Id"]].append(float(r["Counter_Value"]))
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_p8_big" in r["Kernel_Name"]:
            disp[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
clk=sorted(sum(v)/8/disp[d]/1e3 for d,v in val.items() if d in disp and disp[d]>0)
if clk: print(f"   shader clock over {len(clk)} launches: median {clk[len(clk)//2]:.2f} GHz (min {clk[0]:.2f}, max {clk[-1]:.2f})")
PY
done
# the mid-grid kernel (128x128 tiles, the AR decoder's 50- / 100-token steps and the VAE stacks): MID_ABL_NOMFMA one MFMA of twelve per half
# step, _NODMA no DMA after the prologue, _NOBAR no barriers, _NOB the weight operand neither fetched nor read from LDS
export GEMM_ROTATE=8 GEMM_ONLY="t80 qkv,t400 qkv,t1600 qkv,t1600 ffn1,t3200 qkv" GEMM_VARIANTS="28:1"
for v in base mid_NOMFMA mid_NODMA mid_NOBAR mid_NOB base; do
  if [ $v = base ]; then unset ARTALK_LIB; else export ARTALK_LIB=$PWD/tools/build/libartalk_$v.so; fi
  echo "== $v"
  GEMM_ROUNDS=5 timeout -k 10 120 python3 tools/gemm_f16s_bench.py 2>&1 | grep -v amdgpu.ids
done
