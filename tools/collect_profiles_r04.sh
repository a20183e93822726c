#!/bin/bash
# Round-4 profile collection on a 1-GPU MI355X box (run from the repo root through gpurun); every leg is bounded.
# usage: bash tools/collect_profiles_r04.sh [outdir]
O=${1:-gpurun_out/r04}
mkdir -p $O; export TMPDIR=/tmp
R=$PWD
# the driver's line: B = 1 headline (roofline + cpu_baseline + parity) + config 4 at N = 1 (16 rows weak / 128 rows strong, exact and
# decode_precision bf16)
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
: > $O/bench_other_configs.jsonl
for extra in "--opt weight_prefetch=0" "--batch 16 --steps 100" "--batch 16 --topk 50 --temperature 1.0 --steps 100" \
             "--weights fp8 --ctx 2048 --steps 500 --warmup 4" "--ctx 2048" "--batch 64 --steps 50" "--batch 128 --steps 30"; do
  timeout 300 python bench.py --no-cpu-baseline --config4 0 $extra >> $O/bench_other_configs.jsonl 2>> $O/bench_other.err
done
# context prefill by precision: 0 exact, 1 bf16 activations, 2 MX-fp8
for c in 64 512 2048; do for m in 0 1 2; do timeout 200 python tools/prefill_bench.py $c 1 8 $m; done; done 2>&1 | grep "^ctx" > $O/prefill.txt
# kernel-level split of the benchmarked command (B = 1; streamer off: the profiler serialises dispatches) + step anatomy
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/stats -o r04 -- python $R/bench.py --steps 20 --warmup 4 --lean --opt weight_prefetch=0 > $R/$O/stats.log 2>&1
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 4 --lean --opt weight_prefetch=0"
  echo "# (24 frame-steps + the prefills; at::native::* kernels are the synthetic-weight generation in setup, not the path)"; echo
  python tools/rocprof_summary.py $O/stats/r04_results.db 24; } > $O/bench_kernel_stats.md 2>&1
{ echo "# same trace: python tools/step_timeline.py <db>   (B = 1, streamer off)"; echo; python tools/step_timeline.py $O/stats/r04_results.db; } > $O/b1_step_timeline.md 2>&1
rm -rf $O/stats
for b in 16 128; do
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/b$b -o b$b -- python $R/bench.py --batch $b --steps 20 --warmup 4 --lean > $R/$O/b$b.log 2>&1
  cd $R
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --batch $b --steps 20 --warmup 4 --lean"; echo
    python tools/rocprof_summary.py $O/b$b/b${b}_results.db 24; } > $O/bench_b${b}_kernel_stats.md 2>&1
  { echo "# same trace: python tools/step_timeline.py <db>   (B = $b)"; echo; python tools/step_timeline.py $O/b$b/b${b}_results.db; } > $O/b${b}_step_timeline.md 2>&1
  rm -rf $O/b$b
done
# HBM traffic (separate --pmc passes): B = 1 and the config-4 per-GPU shape
bash tools/collect_pmc.sh $O > $O/pmc_b1.log 2>&1
bash tools/collect_pmc.sh $O "--batch 16" > $O/pmc_b16.log 2>&1
# serving latency record (16 rows; with the codec) and 64 rows
timeout 600 python tools/serve_bench.py 64 16 audio > $O/serve_bench.txt 2>&1
timeout 600 python tools/serve_bench.py 256 64 >> $O/serve_bench.txt 2>&1
ls -la $O
