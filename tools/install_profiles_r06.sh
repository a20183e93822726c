#!/bin/bash
# Copy what tools/collect_profiles_r06.sh wrote (default gpurun_out/r06final) into profiles/ under the round-6 names and fold the
# PMC summaries into profiles/hbm_traffic.json (stamped with the source hash of the tree they were measured on).
O=${1:-gpurun_out/r06final}
cp $O/bench_kernel_stats.md profiles/r06_bench_kernel_stats.md; cp $O/bench_b16_kernel_stats.md profiles/r06_bench_b16_kernel_stats.md
cp $O/bench_b128_kernel_stats.md profiles/r06_bench_b128_kernel_stats.md; cp $O/bench_b128_g128off_kernel_stats.md profiles/r06_bench_b128_g128off_kernel_stats.md
cp $O/b1_timeline.md profiles/r06_b1_timeline.md; cp $O/b1_timeline_per_wg.txt profiles/r06_b1_timeline_per_wg.txt; cp $O/b1_timeline_topk50.md profiles/r06_b1_timeline_topk50.md
cp $O/b16_timeline.md profiles/r06_b16_timeline.md; cp $O/b16_timeline_per_wg.txt profiles/r06_b16_timeline_per_wg.txt
cp $O/launch_kinds_b1.json $O/launch_kinds_b16.json profiles/
cp $O/bench_other_configs.jsonl profiles/r06_bench_other_configs.jsonl; cp $O/prefill.txt profiles/r06_prefill.txt
[ -f $O/g128_ubench.txt ] && cut -c1-230 $O/g128_ubench.txt > profiles/r06_g128_ubench.txt
python - <<PY
import json
O='$O/'
recs={}
for tag,f in (('b1','pmc_hbm.json'),('b16','pmc_hbmbatch16.json'),('b128','pmc_hbmbatch128.json'),('b128_g128off','pmc_hbmbatch128optg128=0.json')):
    recs[tag]=json.load(open(O+f)); print(tag, recs[tag]['hbm_read_bytes_per_step'], recs[tag]['hbm_bytes_per_step'])
json.dump(recs, open('profiles/r06_pmc_hbm.json','w'), indent=1)
PY
N="tools/collect_profiles_r06.sh (round 6 final build)"
python tools/pmc_record.py $O/pmc_hbm.json 1 512 bf16 "$N" > /dev/null
python tools/pmc_record.py $O/pmc_hbmbatch16.json 16 512 bf16 "$N" > /dev/null
python tools/pmc_record.py $O/pmc_hbmbatch128.json 128 512 bf16 "$N" > /dev/null
