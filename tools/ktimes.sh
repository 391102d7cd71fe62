#!/bin/bash
# per-kernel average durations of a one-batch-in-flight bench run (GPU box): tools/ktimes.sh [rows] [extra bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-30}; [ $# -gt 0 ] && shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-comm --no-fp32 --no-latency --sustain-seconds 0 --inflight 1 "$@" > /tmp/kt.log 2>&1
python3 - "$N" <<'PY'
import csv, glob, sys, json
f = glob.glob("/tmp/kt/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
d0 = {}
try: d0 = json.loads(open('/tmp/kt.log').read().strip().splitlines()[-1])
except Exception: pass
steps = d0.get('steps', 50) + d0.get('warmup', 5)      # from the bench line: forwarded --steps / --warmup change it
dec = [int(r["Calls"]) for r in rows if r["Name"].startswith("decode_kernel")]
if dec: steps = dec[0]                                  # one decode launch per step: counts the set-up and one-batch loops of bench.py too
tot = 0.0
for r in rows[:int(sys.argv[1])]:
    per_step = float(r["TotalDurationNs"]) / steps / 1e3; tot += per_step
    print("%-64s calls/step %5.1f avg %7.2f us  per step %7.1f us" % (r["Name"][:64], int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, per_step))
print("sum of listed per step: %.1f us" % tot)
try:
    d = json.loads(open("/tmp/kt.log").read().strip().splitlines()[-1]); print("bench ms_per_step", d["ms_per_step"], "value", d["value"])
except Exception as e: print("no bench line", e)
PY
