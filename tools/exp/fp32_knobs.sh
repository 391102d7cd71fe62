mkdir -p gpurun_out/r4b
B="python bench.py --precision fp32 --no-cpu-baseline --no-comm --no-fp32 --no-layers --sustain-seconds 0 --steps 200"
run() { echo "== $1"; env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['one_batch_in_flight'], d['config']['launches_per_step'])"; }
run X=1
run HEP_MBF=none
run HEP_MBF_MAXH=8
run HEP_MBF_MAXH=16
run HEP_MBF_TS=8
run HEP_XBF=0
run HEP_SE_MAXMB=1000
run HEP_CHAIN=0
run HEP_TOWER=0
run HEP_PW_NT2=1
run HEP_PW_NT2=4
run HEP_PW_MT2=0
