#!/bin/bash
# Same-box sweep of HIP runtime switches (read once per process) on the bench step: ms per step and the AR/VAE body time.
# usage: tools/env_sweep.sh out.log "VAR=val VAR2=val" "..." ...     (an empty string = the default environment)
out=$1; shift
mkdir -p "$(dirname "$out")"; : > "$out"
for envs in "$@"; do
  for rep in 1 2; do
    line=$(env $envs python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-alt-mode $BENCH_ARGS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_ms']
print('%.2f ms/step  conv %.2f enc %.2f ada %.2f ar %.2f  rl %s' % (d['ms_per_step'], s['w2v_conv_ms'], s['w2v_encoder_ms'], s['ada_ms'], s['ar_ms'], d['parity']['rounding_level_clips'] if d.get('parity') else None))")
    echo "[${envs:-default}] $BENCH_ARGS rep$rep: $line" | tee -a "$out"
  done
done
