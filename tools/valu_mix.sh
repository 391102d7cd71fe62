#!/bin/bash
# Instruction mix of every kernel, one batch in flight: how much of a wave's issue time is transcendental (swish = v_exp_f32 + v_rcp_f32),
# fma, conversion, and how much of its life it waits.  SQ counters in a pass of their own (no trace domain).
#   usage: tools/valu_mix.sh <tag>   (EXTRA_BENCH="--precision fp32" for fp32 sessions)  ->  gpurun_out/<tag>/valu_mix.json
: "${1:?usage: valu_mix.sh <tag>}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
T="$R/gpurun_out/$1"; mkdir -p "$T"; rm -rf "$T/valu"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $T/valu -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-comm --no-fp32 --no-latency --sustain-seconds 0 $EXTRA_BENCH --inflight 1 > $T/bench_valu.log 2>&1
find $T/valu -name "*agent_info.csv" -delete
python3 - $T <<'PY'
import csv, collections, glob, json, sys
f = glob.glob(sys.argv[1] + "/valu/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_VALU": cnt[k] += 1
out = {}
for k, v in agg.items():
    if "kernel" not in k or not cnt[k]: continue
    n = cnt[k]; valu = v["SQ_INSTS_VALU"] / n; tr = v["SQ_INSTS_VALU_TRANS_F32"] / n
    wc = v["SQ_WAVE_CYCLES"] / n
    # issue-cost weights of MI355X_MICROARCH.md (transcendental 8 cycles, other VALU 4): share of the VALU issue time that is transcendental
    issue = 8 * tr + 4 * (valu - tr)
    out[k] = {"launches": n, "valu_insts_per_launch": round(valu), "trans_f32": round(tr), "fma_f32": round(v["SQ_INSTS_VALU_FMA_F32"] / n), "cvt": round(v["SQ_INSTS_VALU_CVT"] / n),
              "trans_share_of_valu_insts": round(tr / valu, 4) if valu else None, "trans_share_of_valu_issue_time": round(8 * tr / issue, 4) if issue else None,
              "valu_active_share_of_wave_cycles": round(v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], 4) if wc else None,
              "wait_any_share": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 4) if wc else None,
              "wait_inst_any_share": round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 4) if wc else None}
json.dump(out, open(sys.argv[1] + "/valu_mix.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["trans_f32"])[:12]:
    print(k[:60], v["trans_share_of_valu_insts"], v["trans_share_of_valu_issue_time"], v["valu_active_share_of_wave_cycles"], v["wait_any_share"])
PY
