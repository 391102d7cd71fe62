#!/bin/bash
# usage (in the build container): tools/save_profile.sh <run_round tag> <profiles/rNN/prefix>
#   copies the summaries of `tools/run_round.sh <tag> pmc` from gpurun_out/ into profiles/ (tracked)
: "${2:?usage: save_profile.sh <tag> <profiles/rNN/prefix>}"
T=gpurun_out/$1; P=$2
pmc_json() {   # <gpurun_out dir with pmc_fetch/pmc_write> <out.json>
python3 - "$1" "$2" <<'PY'
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
}
for sfx in 1:"" p3:phi3_ f32:fp32_; do
  d=${T}${sfx%%:*}; q=${P}_${sfx##*:}
  [ -d $d/stats ] || continue
  cp $d/stats/*/*_kernel_stats.csv ${q}bench_inflight1_kernel_stats.csv
  cp $d/stats4/*/*_kernel_stats.csv ${q}bench_kernel_stats.csv
  grep '^{' $d/bench_stats.log > ${q}bench_inflight1_line.json
  grep '^{' $d/bench_stats4.log > ${q}bench_line.json
  [ -d $d/pmc_fetch ] && pmc_json $d ${q}pmc_per_kernel.json
done
[ -f $T/bench_full.json ] && cp $T/bench_full.json ${P}_bench_full_line.json
for n in phi3 b64 b32 fp8_b32 2rank_gloo 8rank_gloo; do [ -s $T/bench_$n.json ] && grep '^{' $T/bench_$n.json > ${P}_bench_${n}_line.json; done
[ -f $T/conc.txt ] && grep -v amdgpu.ids $T/conc.txt > ${P}_concurrent_cost_per_launch.txt
[ -f $T/conc_phi3.txt ] && grep -v amdgpu.ids $T/conc_phi3.txt > ${P}_phi3_concurrent_cost_per_launch.txt
[ -f $T/conc_fp32.txt ] && grep -v amdgpu.ids $T/conc_fp32.txt > ${P}_fp32_concurrent_cost_per_launch.txt
for n in fp32 bf16 phi3; do [ -f $T/plan_$n.txt ] && grep -v amdgpu.ids $T/plan_$n.txt > ${P}_plan_per_launch_$n.txt; done
[ -f ${T}1/mfma_util.json ] && cp ${T}1/mfma_util.json ${P}_mfma_util.json
[ -f ${T}1/valu_mix.json ] && cp ${T}1/valu_mix.json ${P}_valu_mix.json
[ -f ${T}f32/valu_mix.json ] && cp ${T}f32/valu_mix.json ${P}_fp32_valu_mix.json
[ -f $T/pytest.log ] && tail -3 $T/pytest.log > ${P}_pytest_gpu_tail.txt
[ -f $T/pytest_fp8_tail.txt ] && cp $T/pytest_fp8_tail.txt ${P}_pytest_gpu_fp8_build_tail.txt
[ -f $T/pytest_poison_tail.txt ] && cp $T/pytest_poison_tail.txt ${P}_pytest_gpu_poison_build_tail.txt
ls -la $(dirname $P)
