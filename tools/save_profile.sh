#!/bin/bash
# usage (in the build container): tools/save_profile.sh <gpurun_out tag> <profiles/rNN/prefix>
T=gpurun_out/$1; P=$2
cp $T/stats/*/*_kernel_stats.csv ${P}_bench_inflight1_kernel_stats.csv
cp $T/stats4/*/*_kernel_stats.csv ${P}_bench_kernel_stats.csv
grep '^{' $T/bench_stats.log > ${P}_bench_inflight1_line.json
grep '^{' $T/bench_stats4.log > ${P}_bench_line.json
[ -f $T/bench_full.json ] && cp $T/bench_full.json ${P}_bench_full_line.json
[ -d $T/pmc_fetch ] && python3 - $T ${P}_pmc_per_kernel.json <<'PY'
import csv, collections, json, glob, sys
out={}
for kind,sub in (("FETCH_SIZE","pmc_fetch"),("WRITE_SIZE","pmc_write")):
    f=glob.glob(sys.argv[1]+"/"+sub+"/*/*_counter_collection.csv")[0]
    agg=collections.defaultdict(lambda:[0.0,0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']!=kind: continue
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        agg[k][0]+=float(r['Counter_Value']); agg[k][1]+=1
    out[kind]={k:{"launches":v[1],"KB_per_launch":round(v[0]/v[1],2)} for k,v in agg.items() if 'kernel' in k}
json.dump(out, open(sys.argv[2],"w"), indent=1, sort_keys=True)
PY
ls -la $(dirname $P)
