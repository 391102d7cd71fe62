mkdir -p gpurun_out/a
python -m pytest tests -m gpu -x -q -s 2>&1 | tail -80 > gpurun_out/a/pytest.log; tail -5 gpurun_out/a/pytest.log
python bench.py > gpurun_out/a/bench_full.json 2> gpurun_out/a/bench_full.err; cut -c1-1500 gpurun_out/a/bench_full.json
python bench.py --gpus 2 --single-device --backend gloo --steps 40 --warmup 5 > gpurun_out/a/bench_2rank_gloo.json 2> gpurun_out/a/bench_2rank_gloo.err; cut -c1-600 gpurun_out/a/bench_2rank_gloo.json; tail -3 gpurun_out/a/bench_2rank_gloo.err
tools/prof_bench.sh a1
