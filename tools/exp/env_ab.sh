#!/bin/bash
# interleaved A/B of plan knobs on the bench workload (GPU box): tools/exp/env_ab.sh <rounds> "X=1" "HEP_MBF_MP=force" ...   (HEP_LIB=... in an entry selects a library)
# EXTRA="--precision fp32" for other bench arguments; prints sustained frames/s (>= 1 s), the one-batch figure, launches per step, then medians
R=$1; shift
BENCH="python bench.py --no-cpu-baseline --no-comm --no-fp32 --no-layers --no-latency --sustain-seconds 1 --steps 200 $EXTRA"
for i in $(seq $R); do
  for e in "$@"; do
    env $e $BENCH 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', d['sustained']['value'], d['one_batch_in_flight']['value'], d['config']['launches_per_step'])"
  done
done | tee /tmp/env_ab.txt
python - <<'PY'
import statistics as st, collections
d = collections.defaultdict(list)
for l in open('/tmp/env_ab.txt'):
    p = l.split(); d[p[0]].append((float(p[1]), float(p[2])))
for k, v in d.items():
    print(f"{k}: sustained median {st.median(x[0] for x in v):.0f} frames/s (min {min(x[0] for x in v):.0f} max {max(x[0] for x in v):.0f}); one batch median {st.median(x[1] for x in v):.0f}")
PY
