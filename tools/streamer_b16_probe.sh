# the weight streamer beside the B = 16 chain (prefetch_batched + the K = 2048 split that lets every launch fit beside it): knob sweep
O=gpurun_out; mkdir -p $O
run() { timeout 200 python bench.py --no-cpu-baseline --config4 0 --lean --batch 16 --steps 100 "$@" 2>>$O/sb16.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); ws = d.get('weight_streamer') or {}
        print('%-90s ms/step %.4f  late %s of %s gave_up %s' % (' '.join(sys.argv[1:]), d['ms_per_step'], ws.get('skipped_late_sample'), (ws.get('segments') or 0) * (ws.get('frames') or 0), ws.get('gave_up')))
" "$@"; }
{
run
run --opt g16_k16=520
B="--opt g16_k16=520 --opt prefetch_batched=1"
run $B
for v in 0 64 256; do run $B --opt prefetch_seg_sleep=$v; done
for v in 8 16; do run $B --opt prefetch_window_mb=$v; done
run $B --opt prefetch_grid=128
run $B --opt prefetch_part_kb=8192
run $B --opt prefetch_depth=8
run $B --opt prefetch_lead=0
} > $O/streamer_b16.txt 2>&1
cat $O/streamer_b16.txt
