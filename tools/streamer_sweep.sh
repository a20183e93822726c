# one-at-a-time sweep of the weight streamer's knobs at B = 1 (bench.py --lean, 200 steps each): run through gpurun
O=gpurun_out/sweep; mkdir -p $O
run() { timeout 200 python bench.py --no-cpu-baseline --config4 0 --lean --steps 200 "$@" 2>>$O/err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); ws = d.get('weight_streamer') or {}
        print('%-50s ms/step %.4f  late %s of %s' % (' '.join(sys.argv[1:]), d['ms_per_step'], ws.get('skipped_late_sample'), (ws.get('segments') or 0) * (ws.get('frames') or 0)))
" "$@"; }
{
run
run
for v in 16 20 28 32; do run --opt prefetch_window_mb=$v; done
for v in 1024 2048 8192; do run --opt prefetch_sub_kb=$v; done
for v in 0 8 32 64; do run --opt prefetch_seg_sleep=$v; done
for v in 128 512; do run --opt prefetch_grid=$v; done
run --opt prefetch_lead=0
run --opt prefetch_skip_late=0
for v in 8 16 32; do run --opt prefetch_depth=$v; done
run --opt nt_decoder=1
run --opt nt_decoder=0
run --opt nt_backbone=0
run
} > $O/sweep.txt 2>&1
cat $O/sweep.txt
