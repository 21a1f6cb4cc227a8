for sk in "0,0" "96,192" "128,256" "256,512" "384,768" "192,768" "64,128"; do
  echo "== splitk $sk"
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-alt-mode --splitk $sk --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stages_ms']['ar_ms'])"
done
