# The ping-pong GEMM kernel inside the real step (ARTALK_PP / ARTALK_PP_MIN switches of engine.hip gemm()): same box, alternating runs.
# Output: gpurun_out/r05/pp_model_ab2.log
O=gpurun_out/r05/pp_model_ab2.log; mkdir -p gpurun_out/r05; : > $O
run() {
  echo "== $1" >> $O
  env $1 timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-mode 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_ms']
print(d['ms_per_step'], 'ms/step; conv', s['w2v_conv_ms'], 'enc', s['w2v_encoder_ms'], 'ada', s['ada_ms'], 'body', s['ar_ms'], 'parity', d['parity']['decision_exact_chunks'], d['parity']['rounding_level_clips'])" >> $O
}
for rep in 1 2 3; do
  run "ARTALK_PP=0"
  run "ARTALK_PP=1 ARTALK_PP_MIN=120"
  run "ARTALK_PP=1 ARTALK_PP_MIN=150"
  run "ARTALK_PP=1 ARTALK_PP_MIN=160"
done
cat $O
