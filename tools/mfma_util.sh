#!/bin/bash
# MFMA utilisation of the kernels that issue MFMAs (north star: "MFMA utilisation on the pointwise layers"):
# SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES per launch, one batch in flight.  Output: gpurun_out/<tag>/mfma/...
: "${1:?usage: mfma_util.sh <tag>}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
T="$R/gpurun_out/$1"; mkdir -p "$T"; rm -rf "$T/mfma"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $T/mfma -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-comm --no-fp32 --no-latency --sustain-seconds 0 $EXTRA_BENCH --inflight 1 > $T/bench_mfma.log 2>&1
find $T/mfma -name "*agent_info.csv" -delete
python3 - $T <<'PY'
import csv, collections, glob, json, sys
f = glob.glob(sys.argv[1] + "/mfma/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
out = {}
for k, v in agg.items():
    if "kernel" not in k or not cnt[k]: continue
    mf, gui = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / cnt[k], v.get("GRBM_GUI_ACTIVE", 0.0) / cnt[k]
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over all 1024 SIMDs
    out[k] = {"launches": cnt[k], "mfma_busy_cycles_per_launch": round(mf), "gui_active_per_launch": round(gui),
              "mfma_util": round(mf / (gui / 8 * 1024), 5) if gui else None}
json.dump(out, open(sys.argv[1] + "/mfma_util.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: v["mfma_util"] for k, v in out.items() if v["mfma_busy_cycles_per_launch"] > 0}, indent=0))
PY
