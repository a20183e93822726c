#!/bin/bash
# kernel trace + step anatomy of a batched step with given engine options: bash tools/b128_trace.sh <batch> <out.md> [--opt ...]
B=$1; OUT=$2; shift 2
R=$PWD; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr_$$ -o t -- python $R/bench.py --batch $B --steps 12 --warmup 4 --lean --no-cpu-baseline --config4 0 "$@" > /tmp/tr_$$.log 2>&1
cd $R
{ echo "# rocprofv3 --kernel-trace -- python bench.py --batch $B --steps 12 --warmup 4 --lean $*"; echo; python tools/step_timeline.py /tmp/tr_$$/t_results.db; } > $OUT 2>&1
rm -rf /tmp/tr_$$
