#!/bin/bash
# Round-3 profile collection on a 1-GPU MI355X box (run from the repo root through gpurun); every leg is bounded.
# usage: bash tools/collect_profiles_r03.sh [outdir]
O=${1:-gpurun_out/r03}
mkdir -p $O; export TMPDIR=/tmp
R=$PWD
# the driver's line: B = 1 headline + the config-4 record (16 rows weak / 128 rows strong through generate_sharded) at N = 1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
: > $O/bench_other_configs.jsonl
for extra in "--opt weight_prefetch=0" "--opt two_token_pass=0" "--batch 16 --steps 100" "--batch 16 --topk 50 --temperature 1.0 --steps 100" \
             "--weights fp8 --steps 300" "--ctx 2048" "--weights fp8 --ctx 2048 --steps 500 --warmup 4" \
             "--topk 50 --temperature 0.9" "--no-graph --steps 100" "--weights fp8 --batch 16 --steps 100" "--batch 32 --steps 50" "--batch 64 --steps 50" \
             "--batch 64 --steps 50 --opt rows64=0"; do
  timeout 300 python bench.py --no-cpu-baseline --config4 0 $extra >> $O/bench_other_configs.jsonl 2>> $O/bench_other.err
done
# context prefill by precision: 0 exact, 1 bf16 activations (LDS-DMA GEMM up to 4096 rows), 2 MX-fp8 weights and activations
for c in 64 512 1024 2048; do for m in 0 1 2; do timeout 200 python tools/prefill_bench.py $c 1 8 $m; done; done 2>&1 | grep "^ctx" > $O/prefill.txt
timeout 200 python tools/prefill_bench.py 512 16 3 >> $O/prefill.txt 2>&1
timeout 200 python tools/prefill_bench.py 512 1 8 1 gemm_dma=0 2>&1 | grep "^ctx" >> $O/prefill.txt
timeout 200 python tools/prefill_bench.py 2048 1 8 1 gemm_dma=0 2>&1 | grep "^ctx" >> $O/prefill.txt
# the 256 x 256 tile (gemm256.h) off: 2048 frames and 16 x 512 frames, bf16 and mxfp8
for m in 1 2; do timeout 200 python tools/prefill_bench.py 2048 1 8 $m gemm_256=0 2>&1 | grep "^ctx" >> $O/prefill.txt; done
for m in 1 2; do timeout 300 python tools/prefill_bench.py 512 16 3 $m 2>&1 | grep "^ctx" >> $O/prefill.txt; timeout 300 python tools/prefill_bench.py 512 16 3 $m gemm_256=0 2>&1 | grep "^ctx" >> $O/prefill.txt; done
timeout 300 python tools/bench_gemm_mx.py 2>&1 | grep "^|" > $O/gemm_mx_microbench.md
# kernel-level split of the benchmarked command (streamer off under the profiler) + the launch-by-launch step anatomy
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/stats -o r03 -- python $R/bench.py --steps 20 --warmup 4 --lean --opt weight_prefetch=0 > $R/$O/stats.log 2>&1
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 4 --lean --opt weight_prefetch=0"
  echo "# (24 frame-steps + the prefills; at::native::* kernels are the synthetic-weight generation in setup, not the path)"; echo
  python tools/rocprof_summary.py $O/stats/r03_results.db 24; } > $O/bench_kernel_stats.md 2>&1
{ echo "# same trace: python tools/step_timeline.py <db>   (B = 1, streamer off)"; echo; python tools/step_timeline.py $O/stats/r03_results.db; } > $O/b1_step_timeline.md 2>&1
rm -rf $O/stats
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/b16 -o b16 -- python $R/bench.py --batch 16 --steps 40 --warmup 4 --lean > $R/$O/b16.log 2>&1
cd $R
{ echo "# BASELINE configs[2]: rocprofv3 --kernel-trace --stats -- python bench.py --batch 16 --steps 40 --warmup 4 --lean"; echo
  python tools/rocprof_summary.py $O/b16/b16_results.db 44; } > $O/bench_b16_kernel_stats.md 2>&1
{ echo "# same trace: python tools/step_timeline.py <db>   (B = 16)"; echo; python tools/step_timeline.py $O/b16/b16_results.db; } > $O/b16_step_timeline.md 2>&1
rm -rf $O/b16
# config 5 with the MX-fp8 prefill: per-kernel split of 2048-frame prefills, then matrix-pipe busy in a separate --pmc pass
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/mx -o mx -- python $R/tools/prefill_bench.py 2048 1 6 2 > $R/$O/mx.log 2>&1
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- python tools/prefill_bench.py 2048 1 6 2   (6 prefills of 2048 frames, prefill_precision = mxfp8)"; echo
  python tools/rocprof_summary.py $O/mx/mx_results.db 6 | grep -v "at::native"; } > $O/prefill2048_mxfp8_kernel_stats.md 2>&1
rm -rf $O/mx
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/$O/mxpmc -o mx -- python $R/tools/prefill_bench.py 2048 1 3 2 > $R/$O/mxpmc.log 2>&1
echo "mx pmc rc=$?"
cd $R
python - $O <<'PY' > $O/config5_pmc_mfma.md 2>> $O/pmc.err
import sqlite3, sys, collections
o = sys.argv[1]
db = sqlite3.connect(f"{o}/mxpmc/mx_results.db")
rows = db.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name").fetchall()
t = collections.defaultdict(dict)
for n, c, k, v in rows:
    t[n][c] = (k, v)
print("# config 5, prefill_precision = mxfp8, 2048 frames: matrix-pipe busy per kernel (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8):")
print("# rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs).  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python tools/prefill_bench.py 2048 1 3 2")
print("| kernel | launches | MFMA busy % of the chip's matrix pipes | GRBM_GUI_ACTIVE cycles / launch |")
print("|---|---|---|---|")
for n, d in sorted(t.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[1])[:12]:
    if "GRBM_GUI_ACTIVE" not in d or not d["GRBM_GUI_ACTIVE"][1] or "at::native" in n: continue
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[1]
    act = d["GRBM_GUI_ACTIVE"][1]
    print(f"| `{n[:80]}` | {d['GRBM_GUI_ACTIVE'][0]} | {100.0 * busy / (128.0 * act):.1f} | {act / d['GRBM_GUI_ACTIVE'][0]:.0f} |")
PY
rm -rf $O/mxpmc
# HBM traffic of the benchmarked command (B = 1, streamer off: counter passes serialise the dispatches)
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=pmc_$(echo $c | tr A-Z a-z)
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/$O/$d -o r03 -- python $R/bench.py --steps 4 --warmup 2 --lean --opt weight_prefetch=0 > $R/$O/$d.log 2>&1
  echo "$c rc=$?"
done
cd $R
python tools/pmc_summary.py $O/pmc_fetch_size/r03_results.db 6 $O/pmc_write_size/r03_results.db > $O/pmc_hbm.json 2> $O/pmc_hbm.err
rm -rf $O/pmc_fetch_size $O/pmc_write_size
timeout 400 python tools/serve_bench.py 64 16 2>&1 | grep -v amdgpu > $O/serve_bench.txt
ls -la $O | head -40; head -c 600 $O/bench.json
# ---- second half of the round: prefill fusions and short-prefill tiles (each script writes gpurun_out/<name>.txt; copied to profiles/r03_*.txt)
bash tools/fuse_probe.sh        > /dev/null 2>&1   # -> profiles/r03_prefill_fused_epilogues.txt   (GEPI_ROPE, MX output of the attention)
bash tools/gu_split_probe.sh    > /dev/null 2>&1   # -> profiles/r03_prefill_gateup_splitk.txt     (split-K gate/up + swiglu_reduce_kernel)
bash tools/skinny_probe.sh      > /dev/null 2>&1   # -> profiles/r03_prefill_skinny_tiles.txt      (64 / 32-row workgroups of the LDS-DMA GEMM)
bash tools/streamer_b16_probe.sh > /dev/null 2>&1  # -> profiles/r03_streamer_b16.txt              (weight streamer beside the B = 16 chain)
bash tools/prefill_profile.sh "2048 1" "2048 2" "512 1" "512 2" > /dev/null 2>&1   # -> profiles/r03_prefill{2048,512}_{bf16,mxfp8}_kernel_stats.md
bash tools/final_check.sh       > /dev/null 2>&1   # -> profiles/r03_bench.json, r03_bench_config5.json, last sections of r03_prefill.txt / r03_serve_bench.txt
