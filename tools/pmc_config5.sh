#!/bin/bash
# config 5 (mxfp8 prefill, 2048 frames) and the bf16-mode prefill: matrix-pipe busy per kernel (one --pmc pass each, kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=/tmp/pmc_c5; mkdir -p $O   # (the result databases are large: scratch, not gpurun_out)
for m in 2 1; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/m$m -o p -- python $R/tools/prefill_bench.py 2048 1 3 $m > $O/m$m.log 2>&1 || echo "pass mode $m failed"
done
cd $R; python tools/pmc_kernels.py $O
