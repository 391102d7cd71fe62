# A/B of phi 3 @ 512 b8 bf16 (BASELINE config 4): bash tools/exp/phi3_ab.sh "ENV=1" ...
B="python bench.py --phi 3 --size 512 --batch 8 --no-cpu-baseline --no-comm --no-fp32 --no-layers --sustain-seconds 0 --steps 100"
for e in "$@"; do
  echo "== $e"; env $e $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['one_batch_in_flight'], d['config']['launches_per_step'])"
done
