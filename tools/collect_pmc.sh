#!/bin/bash
# HBM-traffic counter passes (separate --pmc runs, kernel-trace only) for profiles/hbm_traffic.json.
# rocprofv3 7.2 segfaults in its own counter-collection thread when ~18k dispatches are queued without a host sync
# (24 frame-steps in flight); 6 frame-steps per run are fine, and the per-step traffic does not depend on the count.
# usage: tools/collect_pmc.sh [out dir] [extra bench.py flags, e.g. "--batch 16"]  ->  <out>/pmc_hbm<tag>.json
O=${1:-gpurun_out/final}; X=${2:-}; R=$PWD; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
T=$(echo "$X" | tr -d ' -')
for c in FETCH_SIZE WRITE_SIZE; do
  d=pmc_$(echo $c | tr A-Z a-z | sed s/_size//)$T
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/$O/$d -o r01 -- python $R/bench.py --steps 4 --warmup 2 --lean $X > $R/$O/$d.log 2>&1
  echo "$c rc=$?"
done
cd $R
python tools/pmc_summary.py $O/pmc_fetch$T/r01_results.db 6 $O/pmc_write$T/r01_results.db > $O/pmc_hbm$T.json 2> $O/pmc_hbm$T.err
cat $O/pmc_hbm$T.json | head -40; cat $O/pmc_hbm$T.err
rm -rf $O/pmc_fetch$T $O/pmc_write$T
