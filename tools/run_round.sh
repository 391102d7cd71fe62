#!/bin/bash
# one GPU-box visit: parity tests, the bench line, a 2-rank wiring run, rocprofv3 summaries
T=gpurun_out/${1:-b}; mkdir -p $T
python -m pytest tests -m gpu -q 2>&1 | tail -120 > $T/pytest.log; tail -3 $T/pytest.log
python bench.py > $T/bench_full.json 2> $T/bench_full.err; cut -c1-1200 $T/bench_full.json
python bench.py --gpus 2 --single-device --backend gloo --steps 40 --warmup 5 > $T/bench_2rank_gloo.json 2> $T/bench_2rank_gloo.err; cut -c1-300 $T/bench_2rank_gloo.json
tools/prof_bench.sh ${1:-b}1 $2
