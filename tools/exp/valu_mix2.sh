R="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
T="$R/gpurun_out/r4v2"; mkdir -p "$T"; rm -rf "$T/valu"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_LDS --output-format csv -d $T/valu -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-comm --no-fp32 --no-latency --sustain-seconds 0 $EXTRA_BENCH --inflight 1 > $T/bench_valu.log 2>&1
python3 - $T <<'PY'
import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + "/valu/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_VALU": cnt[k] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"]):
    if "kernel" not in k: continue
    n = cnt[k]; tot = v["SQ_INSTS_VALU"]
    named = sum(v[c] for c in ("SQ_INSTS_VALU_INT32","SQ_INSTS_VALU_MUL_F32","SQ_INSTS_VALU_ADD_F32","SQ_INSTS_VALU_FMA_F32","SQ_INSTS_VALU_TRANS_F32","SQ_INSTS_VALU_CVT"))
    print(k[:52].ljust(52), "valu/launch %9d" % (tot/n), " ".join("%s %.2f" % (c.replace("SQ_INSTS_VALU_",""), v[c]/tot) for c in ("SQ_INSTS_VALU_INT32","SQ_INSTS_VALU_MUL_F32","SQ_INSTS_VALU_ADD_F32","SQ_INSTS_VALU_FMA_F32","SQ_INSTS_VALU_TRANS_F32","SQ_INSTS_VALU_CVT")), "other %.2f" % (1-named/tot), "lds/valu %.2f" % (v["SQ_INSTS_LDS"]/tot))
PY
