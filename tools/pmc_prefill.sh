#!/bin/bash
# per-kernel hardware counters of one bf16-mode prefill (separate --pmc passes): tools/pmc_prefill.sh <ctx> <outdir>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; CTX=${1:-2048}; O=$R/${2:-gpurun_out/pmc_prefill}
mkdir -p $O
for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_WAIT_ANY" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"; do
  d=$(echo $c | cut -d' ' -f1 | tr A-Z a-z)
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O/$d -o p -- python $R/tools/prefill_bench.py $CTX 1 2 1 > $O/$d.log 2>&1 || echo "pass $c failed"
done
ls $O
