# A/B + knock-out timing of the decode chain (results of dbg_skip runs are WRONG by construction: timing only)
O=gpurun_out/ablate; mkdir -p $O
run() { timeout 200 python bench.py --no-cpu-baseline --config4 0 --lean "$@" 2>>$O/err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-60s ms/step %.4f  frames/s %.1f' % (' '.join(sys.argv[1:]), d['ms_per_step'], d['value']))
" "$@"; }
{
run --steps 200
run --steps 200 --opt dbg_skip=128
run --steps 200
run --steps 200 --opt dbg_skip=128
run --steps 200 --kv-dtype bf16
run --steps 200 --kv-dtype bf16 --opt dbg_skip=128
run --steps 200 --opt weight_prefetch=0
run --steps 200 --opt weight_prefetch=0 --opt dbg_skip=128
} > $O/ablate5.txt 2>&1
cat $O/ablate5.txt
