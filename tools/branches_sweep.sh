# Clip groups of the AR/VAE body (1 / 2 / 3 / 4 concurrent graphs): body time of the headline step per group count, same box.
for b in 2 3 4 1 2 3; do
  timeout -k 10 200 python3 bench.py --steps 8 --warmup 3 --branches $b --no-cpu-baseline --no-alt-mode 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_ms']
print('branches $b:', d['ms_per_step'], 'ms/step; body', s['ar_ms'], 'parity', d['parity']['decision_exact_chunks'], d['parity']['rounding_level_clips'])"
done
