#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_bench.sh <tag> [pmc]
#   -> gpurun_out/<tag>/{stats,stats4[,pmc_fetch,pmc_write]} + bench lines; tools/save_profile.sh copies the
#      summaries into profiles/.  Counters are collected in their own passes (never with a trace domain).
: "${1:?usage: prof_bench.sh <tag> [pmc] [extra bench.py args...]}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
T="$R/gpurun_out/$1"
rm -rf "$T"; mkdir -p "$T"
# one batch in flight: per-launch durations comparable with bench.py's roofline block
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $T/stats -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-comm --no-fp32 --no-latency --sustain-seconds 0 $EXTRA_BENCH --inflight 1 > $T/bench_stats.log 2>&1
tail -1 $T/bench_stats.log | cut -c1-400
# the default command (4 batches in flight): launches of different batches overlap, durations stretch
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $T/stats4 -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-comm --no-fp32 --no-latency --sustain-seconds 0 $EXTRA_BENCH > $T/bench_stats4.log 2>&1
tail -1 $T/bench_stats4.log | cut -c1-400
if [ "$2" = "pmc" ]; then
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $T/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-comm --no-fp32 --no-latency --sustain-seconds 0 $EXTRA_BENCH --inflight 1 > $T/bench_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $T/pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-comm --no-fp32 --no-latency --sustain-seconds 0 $EXTRA_BENCH --inflight 1 > $T/bench_write.log 2>&1
fi
# keep the merge-back small: only the stats summaries and counter tables travel
find $T -name "*kernel_trace.csv" -delete
find $T -name "*agent_info.csv" -delete
du -sh $T
