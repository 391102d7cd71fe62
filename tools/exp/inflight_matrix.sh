#!/bin/bash
# frames/s against batches in flight for plan alternatives (round 5: do the narrow fused launches want more streams?)
B="python bench.py --no-cpu-baseline --no-comm --no-fp32 --no-layers --sustain-seconds 0 --steps 200"
for e in "X=0" "HEP_LATE=1" "HEP_LATE=1 HEP_HEADS_FUSED=1" "HEP_LATE=1 GPU_MAX_HW_QUEUES=8"; do
  for n in 3 4 6 8; do
    echo -n "$e inflight $n: "; env $e $B --inflight $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['one_batch_in_flight']['value'])"
  done
done
