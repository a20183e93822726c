#!/bin/bash
# Round-2 profile collection on a 1-GPU MI355X box (run from the repo root through gpurun); every leg is bounded.
# usage: bash tools/collect_profiles_r02.sh [outdir]
O=${1:-gpurun_out/r02}
mkdir -p $O; export TMPDIR=/tmp
R=$PWD
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err
: > $O/bench_other_configs.jsonl
for extra in "--opt weight_prefetch=0" "--batch 16 --steps 100" "--batch 16 --topk 50 --temperature 1.0 --steps 100" \
             "--weights fp8 --steps 300" "--ctx 2048" "--weights fp8 --ctx 2048 --steps 500 --warmup 4" \
             "--topk 50 --temperature 0.9" "--no-graph --steps 100" "--weights fp8 --batch 16 --steps 100"; do
  timeout 300 python bench.py --no-cpu-baseline $extra >> $O/bench_other_configs.jsonl 2>> $O/bench_other.err
done
for c in 512 1024 2048; do timeout 200 python tools/prefill_bench.py $c 1 5; done > $O/prefill.txt 2>&1
timeout 200 python tools/prefill_bench.py 512 16 3 >> $O/prefill.txt 2>&1
# kernel-level split of the benchmarked command (streamer off under the profiler: it is one persistent launch that
# would dwarf every row; the bench line above is the un-profiled number)
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/stats -o r02 -- python $R/bench.py --steps 20 --warmup 4 --lean --opt weight_prefetch=0 > $R/$O/stats.log 2>&1
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 4 --lean --opt weight_prefetch=0"
  echo "# (24 frame-steps + the prefills; at::native::* kernels are the synthetic-weight generation in setup, not the path)"; echo
  python tools/rocprof_summary.py $O/stats/r02_results.db 24; } > $O/bench_kernel_stats.md 2>&1
rm -rf $O/stats
# batch of 16 (BASELINE configs[2]): kernel split
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/b16 -o b16 -- python $R/bench.py --batch 16 --steps 40 --warmup 4 --lean > $R/$O/b16.log 2>&1
cd $R
{ echo "# BASELINE configs[2]: rocprofv3 --kernel-trace --stats -- python bench.py --batch 16 --steps 40 --warmup 4 --lean"; echo
  python tools/rocprof_summary.py $O/b16/b16_results.db 44; } > $O/bench_b16_kernel_stats.md 2>&1
rm -rf $O/b16
# prefill kernels, prefill_precision = bf16, 2048 frames: per-kernel split and hardware counters (separate --pmc passes)
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/pf -o pf -- python $R/tools/prefill_bench.py 2048 1 6 1 > $R/$O/pf.log 2>&1
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- python tools/prefill_bench.py 2048 1 6 1   (6 prefills of 2048 frames, prefill_precision = bf16)"; echo
  python tools/rocprof_summary.py $O/pf/pf_results.db 6; } > $O/prefill2048_bf16_kernel_stats.md 2>&1
rm -rf $O/pf
GRAFT_REPO_ROOT=$R bash tools/pmc_prefill.sh 2048 $O/pmc_prefill > /dev/null 2>&1
python tools/pmc_kernels.py $O/pmc_prefill > $O/prefill2048_bf16_pmc.txt 2>&1
rm -rf $O/pmc_prefill
# config 5 (fp8 weights, 2048-frame prefill, 500 frames): kernel split, then HBM bytes + matrix-pipe busy in separate --pmc passes
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/c5 -o c5 -- python $R/bench.py --weights fp8 --ctx 2048 --steps 40 --warmup 4 --lean --opt weight_prefetch=0 > $R/$O/c5.log 2>&1
cd $R
{ echo "# BASELINE configs[4]: rocprofv3 --kernel-trace --stats -- python bench.py --weights fp8 --ctx 2048 --steps 40 --warmup 4 --lean --opt weight_prefetch=0"; echo
  python tools/rocprof_summary.py $O/c5/c5_results.db 44; } > $O/config5_kernel_stats.md 2>&1
rm -rf $O/c5
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=c5pmc_$(echo $c | cut -d' ' -f1 | tr A-Z a-z)
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/$O/$d -o c5 -- python $R/bench.py --weights fp8 --ctx 2048 --steps 4 --warmup 2 --lean --opt weight_prefetch=0 > $R/$O/$d.log 2>&1
  echo "$c rc=$?"
done
cd $R
python tools/pmc_summary.py $O/c5pmc_fetch_size/c5_results.db 6 $O/c5pmc_write_size/c5_results.db > $O/config5_pmc_hbm.json 2> $O/config5_pmc.err
python - $O <<'PY' > $O/config5_pmc_mfma.md 2>> $O/config5_pmc.err
import sqlite3, sys, collections
o = sys.argv[1]
db = sqlite3.connect(f"{o}/c5pmc_sq_valu_mfma_busy_cycles/c5_results.db")
rows = db.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name").fetchall()
t = collections.defaultdict(dict)
for n, c, k, v in rows:
    t[n][c] = (k, v)
print("# config 5: matrix-pipe busy per kernel (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs), prefill + decode kernels")
print("| kernel | launches | MFMA busy % of the chip's matrix pipes | GRBM_GUI_ACTIVE cycles / launch |")
print("|---|---|---|---|")
for n, d in sorted(t.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[1])[:14]:
    if "GRBM_GUI_ACTIVE" not in d or not d["GRBM_GUI_ACTIVE"][1]: continue
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[1]
    act = d["GRBM_GUI_ACTIVE"][1]
    print(f"| `{n[:80]}` | {d['GRBM_GUI_ACTIVE'][0]} | {100.0 * busy / (128.0 * act):.1f} | {act / d['GRBM_GUI_ACTIVE'][0]:.0f} |")
PY
rm -rf $O/c5pmc_*
# HBM traffic of the benchmarked command (B = 1, streamer off: counter passes serialise the dispatches)
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=pmc_$(echo $c | tr A-Z a-z)
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/$O/$d -o r02 -- python $R/bench.py --steps 4 --warmup 2 --lean --opt weight_prefetch=0 > $R/$O/$d.log 2>&1
done
cd $R
python tools/pmc_summary.py $O/pmc_fetch_size/r02_results.db 6 $O/pmc_write_size/r02_results.db > $O/pmc_hbm.json 2> $O/pmc_hbm.err
rm -rf $O/pmc_fetch_size $O/pmc_write_size
timeout 300 python tools/streamer_probe.py > $O/streamer_probe.txt 2>&1
for p in 1 220 512; do echo "== pool_mb=$p nt=0"; timeout 200 python tools/bench_gemv.py pool_mb=$p nt=0 2>/dev/null | tail -10; done > $O/gemv_pool_microbench.txt 2>&1
# "next" rows: Mimi decode (f-2), continuous batching (f-4), the end-to-end streaming loop; the launch-chain micro-benchmark
timeout 300 python tools/mimi_bench.py 2>&1 | grep -v "amdgpu\|rope_param" > $O/mimi_bench.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/mimi -o mimi -- python $R/tools/mimi_bench.py 200 > /dev/null 2>&1
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- python tools/mimi_bench.py 200   (6 decodes of 200 frames = 16 s of audio each, kyutai/mimi shape, fp32)"; echo
  python tools/rocprof_summary.py $O/mimi/mimi_results.db 6 2>/dev/null | grep -v "at::native" | head -30; } > $O/mimi_kernel_stats.md
rm -rf $O/mimi
timeout 400 python tools/serve_bench.py 64 16 2>&1 | grep -v amdgpu > $O/serve_bench.txt
timeout 300 python tools/stream_demo.py 200 512 2>&1 | grep -v amdgpu | tail -1 > $O/stream_demo.txt
(cd tools/ubench && hipcc --offload-arch=gfx950 -O3 flagchain.hip -o flagchain 2>/dev/null; timeout 120 ./flagchain) > $O/flagchain_ubench.txt 2>&1
# Mimi streaming calls by the skinny-GEMM row threshold; a short one-shot decode per kernel; short-context prefill by the K-split cap;
# phase knock-outs of the square-tile prefill GEMM
for sk in 16 4 0 64; do CSM_MIMI_SKINNY=$sk timeout 200 python tools/mimi_stream_probe.py 2>&1 | grep -v amdgpu | tail -5; done > $O/mimi_stream_probe.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/m25 -o m -- python $R/tools/mimi_short_profile.py > /dev/null 2>&1
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- python tools/mimi_short_profile.py   (45 one-shot decodes of 25 frames, kyutai/mimi shape)"; echo
  python tools/rocprof_summary.py $O/m25/m_results.db 45 2>/dev/null | head -30; } > $O/mimi_short_decode_kernel_stats.md
rm -rf $O/m25
timeout 600 python tools/splitk_sweep.py 2>&1 | grep "^ctx" > $O/prefill_splitk_sweep.txt
for sp in 1 0; do echo "== CSM_MIMI_SPLITK=$sp"; for T in 25 50 100; do CSM_MIMI_SPLITK=$sp timeout 200 python tools/mimi_short_profile.py $T 2>&1 | grep "per decode"; done
  CSM_MIMI_SPLITK=$sp timeout 300 python tools/mimi_bench.py 2>&1 | grep "frames =\|streaming"; done > $O/mimi_splitk.txt
for sk in 4 16 32 64; do echo "== CSM_MIMI_SKINNY=$sk (K split on)"; CSM_MIMI_SKINNY=$sk timeout 200 python tools/mimi_stream_probe.py 2>&1 | grep -v amdgpu | tail -5
  for T in 6 12 16 25 32; do CSM_MIMI_SKINNY=$sk timeout 200 python tools/mimi_short_profile.py $T 2>&1 | grep "per decode"; done; done > $O/mimi_threshold.txt
(cd tools/ubench && hipcc --offload-arch=gfx950 -O3 gemmphase.hip -o gemmphase 2>/dev/null; timeout 120 ./gemmphase) > $O/gemmphase_ubench.txt 2>&1
ls -la $O | head -40; cat $O/bench.json | head -5
