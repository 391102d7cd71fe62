# A/B of fp32 sessions: bash tools/exp/fp32_ab.sh "ENV=1" "ENV2=x" ...   (each argument is one environment setting; "X=1" = default)
B="python bench.py --precision ${PREC:-fp32} --no-cpu-baseline --no-comm --no-fp32 --no-layers --sustain-seconds 0 --steps 200"
for e in "$@"; do
  echo "== $e"; env $e $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['one_batch_in_flight'], d['config']['launches_per_step'])"
done
