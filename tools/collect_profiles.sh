#!/bin/bash
# Collects everything profiles/ holds for a round, on a 1-GPU MI355X box (run from the repo root through gpurun).
# usage: bash tools/collect_profiles.sh [outdir]      (every leg is bounded by `timeout`)
# The summaries are produced on the box from the rocpd databases with tools/rocprof_summary.py / tools/pmc_summary.py.
O=${1:-gpurun_out/final}
mkdir -p $O; export TMPDIR=/tmp
R=$PWD
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
: > $O/bench_other.jsonl
for extra in "--batch 16 --steps 100" "--weights fp8 --steps 400" "--ctx 2048" "--batch 16 --topk 50 --temperature 0.9 --steps 100" \
             "--topk 50 --temperature 0.9" "--no-graph --steps 100" "--kv-dtype bf16" "--weights fp8 --batch 16 --steps 100"; do
  timeout 300 python bench.py --no-cpu-baseline $extra >> $O/bench_other.jsonl 2>> $O/bench_other.err
done
timeout 300 python tools/bench_gemv.py 1 16 > $O/gemv_microbench.md 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/stats -o r01 -- python $R/bench.py --steps 20 --warmup 4 --lean > $R/$O/stats.log 2>&1
cd $R
# summarise on the box (the three rocpd databases together are close to gpurun's 64 MiB return limit)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 4 --lean"
  echo "# (24 frame-steps + two prefills; the at::native::* kernels are the synthetic-weight generation in setup, not the path)"; echo
  python tools/rocprof_summary.py $O/stats/r01_results.db 24; } > $O/kernel_stats.md 2>&1
bash tools/collect_pmc.sh $O > $O/pmc.log 2>&1
ls -la $O $O/stats $O/pmc_fetch | head -40
tail -3 $O/pytest_gpu.log; cat $O/bench_default.json
