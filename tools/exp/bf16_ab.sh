# A/B of bf16 sessions at a batch size: BATCH=32 bash tools/exp/bf16_ab.sh "ENV=1" ...
B="python bench.py --batch ${BATCH:-16} --no-cpu-baseline --no-comm --no-fp32 --no-layers --sustain-seconds 0 --steps 200"
for e in "$@"; do
  echo "== batch ${BATCH:-16} $e"; env $e $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['one_batch_in_flight'], d['config']['launches_per_step'])"
done
