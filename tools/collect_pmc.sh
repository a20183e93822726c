#!/bin/bash
# HBM-traffic counter passes (separate --pmc runs, kernel-trace only) for profiles/hbm_traffic.json.
# rocprofv3 7.2 segfaults in its own counter-collection thread when ~18k dispatches are queued without a host sync
# (24 frame-steps in flight); 6 frame-steps per run are fine, and the per-step traffic does not depend on the count.
O=${1:-gpurun_out/final}; R=$PWD; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=pmc_$(echo $c | tr A-Z a-z | sed s/_size//)
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/$O/$d -o r01 -- python $R/bench.py --steps 4 --warmup 2 --lean > $R/$O/$d.log 2>&1
  echo "$c rc=$?"
done
cd $R
python tools/pmc_summary.py $O/pmc_fetch/r01_results.db 6 $O/pmc_write/r01_results.db > $O/pmc_hbm.json 2> $O/pmc_hbm.err
cat $O/pmc_hbm.json | head -40; cat $O/pmc_hbm.err
rm -rf $O/pmc_fetch $O/pmc_write
