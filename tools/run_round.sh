#!/bin/bash
# one GPU-box visit: parity tests, the bench line, a 2-rank wiring run, rocprofv3 summaries (phi 0 @ 256 b16 and phi 3 @ 512 b8)
# usage: tools/run_round.sh <tag> [pmc]
: "${1:?usage: run_round.sh <tag> [pmc]}"
T=gpurun_out/$1; mkdir -p $T
python -m pytest tests -m gpu -q 2>&1 | tail -120 > $T/pytest.log; tail -3 $T/pytest.log
python bench.py > $T/bench_full.json 2> $T/bench_full.err; cut -c1-600 $T/bench_full.json
python bench.py --gpus 2 --single-device --backend gloo --steps 40 --warmup 5 --no-fp32 --no-layers > $T/bench_2rank_gloo.json 2> $T/bench_2rank_gloo.err; cut -c1-300 $T/bench_2rank_gloo.json
# eight ranks on the box's one GPU over gloo: the wiring of configs 3 / 5 (scatter of 128 uint8 frames, gather of the rows) before the driver's real 8-GPU run
python bench.py --gpus 8 --single-device --backend gloo --steps 8 --warmup 2 --inflight 2 --no-fp32 --no-layers --sustain-seconds 0 > $T/bench_8rank_gloo.json 2> $T/bench_8rank_gloo.err; cut -c1-300 $T/bench_8rank_gloo.json
python bench.py --phi 3 --size 512 --batch 8 --no-cpu-baseline --no-comm --no-latency > $T/bench_phi3.json 2> $T/bench_phi3.err; cut -c1-300 $T/bench_phi3.json
python bench.py --batch 64 --no-cpu-baseline --no-comm --no-fp32 --no-layers --no-latency > $T/bench_b64.json 2> /dev/null; cut -c1-200 $T/bench_b64.json
# the opt-in fp8 build (make -C hmd_ego_pose_amd/csrc fp8, done before the snapshot): its tests and its batch-32 line next to bf16's
if [ -f hmd_ego_pose_amd/libhep_fp8.so ]; then
  HEP_LIB=$PWD/hmd_ego_pose_amd/libhep_fp8.so python -m pytest tests/test_gpu_parity.py -q -m gpu -k fp8 2>&1 | tail -2 > $T/pytest_fp8_tail.txt; cat $T/pytest_fp8_tail.txt
  HEP_LIB=$PWD/hmd_ego_pose_amd/libhep_fp8.so python bench.py --precision fp8 --batch 32 --no-cpu-baseline --no-comm --no-fp32 --no-layers --no-latency > $T/bench_fp8_b32.json 2> /dev/null; cut -c1-200 $T/bench_fp8_b32.json
fi
# the sanitizer build (make -C hmd_ego_pose_amd/csrc poison): the same GPU suite with NaN-poisoned LDS
if [ -f hmd_ego_pose_amd/libhep_poison.so ]; then
  HEP_LIB=$PWD/hmd_ego_pose_amd/libhep_poison.so python -m pytest tests -q -m gpu 2>&1 | tail -2 > $T/pytest_poison_tail.txt; cat $T/pytest_poison_tail.txt
fi
python bench.py --batch 32 --no-cpu-baseline --no-comm --no-fp32 --no-layers --no-latency > $T/bench_b32.json 2> /dev/null; cut -c1-200 $T/bench_b32.json
python tools/conc_profile.py > $T/conc.txt 2>&1
tools/prof_bench.sh ${1}1 $2
EXTRA_BENCH="--phi 3 --size 512 --batch 8" tools/prof_bench.sh ${1}p3 $2
python tools/conc_profile.py 8 bf16 3 512 > $T/conc_phi3.txt 2>&1
# fp32 sessions - the precision inside the 0.1 mm ADD bound: kernel stats, PMC, concurrent cost, per-launch plan
EXTRA_BENCH="--precision fp32" tools/prof_bench.sh ${1}f32 $2
python tools/conc_profile.py 16 fp32 > $T/conc_fp32.txt 2>&1
python tools/plan.py 0 256 16 fp32 > $T/plan_fp32.txt 2>&1
python tools/plan.py 0 256 16 bf16 > $T/plan_bf16.txt 2>&1
python tools/plan.py 3 512 8 bf16 > $T/plan_phi3.txt 2>&1
tools/mfma_util.sh ${1}1
# SQ instruction counters per kernel (VALU instructions per launch, transcendental share, VALU-active / waiting share of the wave cycles)
tools/valu_mix.sh ${1}1
EXTRA_BENCH="--precision fp32" tools/valu_mix.sh ${1}f32
