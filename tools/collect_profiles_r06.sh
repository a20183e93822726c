#!/bin/bash
# Round-6 profile collection on a 1-GPU MI355X box (run from the repo root through gpurun); every leg is bounded.
# usage: bash tools/collect_profiles_r06.sh [outdir]
O=${1:-gpurun_out/r06final}
mkdir -p $O; export TMPDIR=/tmp
R=$PWD
# kernel-level split of the benchmarked command (the profiler serialises the streamer's queue against the replaying graph: these
# durations are the streamer-off ones; the in-step timeline below is the streamer-on record)
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/stats -o r06 -- python $R/bench.py --steps 20 --warmup 4 --lean > $R/$O/stats.log 2>&1
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 4 --lean"
  echo "# (24 frame-steps + the prefills; at::native::* kernels are the synthetic-weight generation in setup, not the path)"; echo
  python tools/rocprof_summary.py $O/stats/r06_results.db 24; } > $O/bench_kernel_stats.md 2>&1
{ echo "# same trace: python tools/step_timeline.py <db>   (B = 1, under the profiler)"; echo; python tools/step_timeline.py $O/stats/r06_results.db; } > $O/b1_step_timeline_rocprof.md 2>&1
rm -rf $O/stats
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/b16 -o b16 -- python $R/bench.py --batch 16 --steps 20 --warmup 4 --lean > $R/$O/b16.log 2>&1
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --batch 16 --steps 20 --warmup 4 --lean"; echo
  python tools/rocprof_summary.py $O/b16/b16_results.db 24; } > $O/bench_b16_kernel_stats.md 2>&1
rm -rf $O/b16
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/b128 -o b128 -- python $R/bench.py --batch 128 --steps 10 --warmup 3 --lean > $R/$O/b128.log 2>&1
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --batch 128 --steps 10 --warmup 3 --lean   (13 frame-steps + the prefill; gemm128_kernel = the FFN launches, csrc/gemm128.h)"; echo
  python tools/rocprof_summary.py $O/b128/b128_results.db 13; } > $O/bench_b128_kernel_stats.md 2>&1
rm -rf $O/b128
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/b128o -o b128o -- python $R/bench.py --batch 128 --steps 10 --warmup 3 --lean --opt g128=0 > $R/$O/b128o.log 2>&1
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --batch 128 --steps 10 --warmup 3 --lean --opt g128=0   (the FFN launches on gemm32_kernel: the A side)"; echo
  python tools/rocprof_summary.py $O/b128o/b128o_results.db 13; } > $O/bench_b128_g128off_kernel_stats.md 2>&1
rm -rf $O/b128o
# gemm128_kernel alone, with its chunk time stamps (tools/ubench/g128_bench.hip; -DCSM_G128_VARIANT=64 build)
if [ -x tools/ubench/bin/g128_bench_h1_0 ]; then
  { for b in tools/ubench/bin/g128_bench_h1_0 tools/ubench/bin/g128_bench_h1_64; do for k in gateup down; do timeout 60 $b $k 300 | head -4 | cut -c1-260; done; done
    timeout 60 tools/ubench/bin/g128_bench_h1_0 gateup 300 64; timeout 60 tools/ubench/bin/g128_bench_h1_0 down 300 64; } > $O/g128_ubench.txt 2>&1
fi
# HBM traffic (separate --pmc passes): B = 1 and the config-4 per-GPU shape
bash tools/collect_pmc.sh $O > $O/pmc_b1.log 2>&1
bash tools/collect_pmc.sh $O "--batch 16" > $O/pmc_b16.log 2>&1
bash tools/collect_pmc.sh $O "--batch 128" > $O/pmc_b128.log 2>&1
bash tools/collect_pmc.sh $O "--batch 128 --opt g128=0" > $O/pmc_b128off.log 2>&1
# in-step timelines (streamer on) + the per-launch-kind tables the bench line attaches
CSM_TL_LIB=$R/csm-hf_amd/libcsm_hip_timeline.so timeout 600 python tools/b1_timeline.py --md $O/b1_timeline.md --json $O/launch_kinds_b1.json > /dev/null 2>&1
CSM_TL_LIB=$R/csm-hf_amd/libcsm_hip_timeline.so timeout 600 python tools/b1_timeline.py --topk 50 --md $O/b1_timeline_topk50.md > /dev/null 2>&1
CSM_TL_LIB=$R/csm-hf_amd/libcsm_hip_timeline.so timeout 600 python tools/b1_timeline.py --batch 16 --md $O/b16_timeline.md --json $O/launch_kinds_b16.json > /dev/null 2>&1
# other configurations through the bench
: > $O/bench_other_configs.jsonl
for extra in "--opt weight_prefetch=0" "--topk 50 --temperature 0.9" "--topk 50 --temperature 0.9 --opt fuse_sample=0" "--batch 16 --steps 100" "--batch 16 --steps 100 --opt g16_kfast=0" "--batch 16 --topk 50 --temperature 1.0 --steps 100" \
             "--weights fp8 --ctx 2048 --steps 500 --warmup 4" "--ctx 2048" "--kv-dtype bf16" "--batch 64 --steps 50" "--batch 96 --steps 30" "--batch 128 --steps 30" "--batch 128 --steps 30 --opt g128=0" "--batch 128 --steps 30 --kv-dtype bf16"; do
  timeout 300 python bench.py --no-cpu-baseline --config4 0 $extra >> $O/bench_other_configs.jsonl 2>> $O/bench_other.err
done
for c in 64 512 2048; do for m in 0 1 2; do timeout 200 python tools/prefill_bench.py $c 1 8 $m; done; done 2>&1 | grep "^ctx" > $O/prefill.txt
ls -la $O
