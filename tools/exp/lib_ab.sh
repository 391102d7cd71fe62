#!/bin/bash
# interleaved A/B of two builds of the library on the bench workload (GPU box):
#   tools/exp/lib_ab.sh <libA.so> <libB.so> [rounds] [extra bench.py args, e.g. --precision fp32]
# prints per run: sustained frames/s (four batches in flight, >= 1 s), the one-batch figure, launches per step; then the medians
A=$1; B=$2; R=${3:-3}; shift 3 2>/dev/null
BENCH="python bench.py --no-cpu-baseline --no-comm --no-fp32 --no-layers --no-latency --sustain-seconds 1 --steps 200 $*"
for i in $(seq $R); do
  for L in "$A" "$B"; do
    HEP_LIB=$PWD/$L $BENCH 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', d['sustained']['value'], d['one_batch_in_flight']['value'], d['config']['launches_per_step'])"
  done
done | tee /tmp/lib_ab.txt
python - <<'PY'
import statistics as st, collections
d = collections.defaultdict(list)
for l in open('/tmp/lib_ab.txt'):
    p = l.split(); d[p[0]].append((float(p[1]), float(p[2])))
for k, v in d.items():
    print(f"{k}: sustained median {st.median(x[0] for x in v):.0f} frames/s (min {min(x[0] for x in v):.0f} max {max(x[0] for x in v):.0f}); one batch median {st.median(x[1] for x in v):.0f}")
PY
