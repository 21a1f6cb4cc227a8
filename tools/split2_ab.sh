# Two-launch answer to the round quantisation of the encoder's FFN-in GEMM (ARTALK_P8_SPLIT2), same box, alternating runs.
O=gpurun_out/r05/split2_ab.log; mkdir -p gpurun_out/r05; : > $O
run() {
  echo "== $1" >> $O
  env $1 timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-mode 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_ms']; r=d['roofline']
print(d['ms_per_step'], 'ms/step; conv', s['w2v_conv_ms'], 'enc', s['w2v_encoder_ms'], 'ada', s['ada_ms'], 'body', s['ar_ms'], '| dominant GEMM', r['launches'], 'launches', r['share_of_step_ms'], 'ms frac', r['frac'], '| parity', d['parity']['decision_exact_chunks'], d['parity']['rounding_level_clips'])" >> $O
}
for rep in 1 2 3; do
  run "ARTALK_P8_SPLIT2=0"
  run "ARTALK_P8_SPLIT2=1"
done
cat $O
