# Same-box A/B of two builds of the library on the headline step: bench.py alternately with ARTALK_LIB=A and =B (N rounds), medians of
# ms_per_step and of the stage times.  usage: tools/ab_bench.sh path/to/libA.so path/to/libB.so [rounds] [extra bench args]
A=$1; B=$2; N=${3:-3}; shift 3 2>/dev/null
for i in $(seq $N); do
  for v in A B; do
    if [ $v = A ]; then L=$A; else L=$B; fi
    ARTALK_LIB=$L timeout -k 10 150 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-mode "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('$v', d['ms_per_step'], s['w2v_conv_ms'], s['w2v_encoder_ms'], s['ada_ms'], s['ar_ms'])"
  done
done | tee /tmp/ab.txt
python3 - <<'PY'
import collections
r=collections.defaultdict(list)
for l in open('/tmp/ab.txt'):
    p=l.split(); r[p[0]].append([float(x) for x in p[1:]])
med=lambda a: sorted(a)[len(a)//2]
for v in sorted(r):
    cols=list(zip(*r[v]))
    print(v, 'median ms_per_step %.2f  conv %.2f  encoder %.2f  ada %.2f  ar %.2f' % tuple(med(c) for c in cols))
PY
