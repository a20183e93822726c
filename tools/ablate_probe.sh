# A/B + knock-out timing of the decode chain (results of dbg_skip runs are WRONG by construction: timing only)
O=gpurun_out/ablate; mkdir -p $O
run() { timeout 200 python bench.py --no-cpu-baseline --config4 0 --lean "$@" 2>>$O/err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-60s ms/step %.4f  frames/s %.1f  checksum %s' % (' '.join(sys.argv[1:]), d['ms_per_step'], d['value'], d.get('tokens_checksum_per_rank')))
" "$@"; }
{
run --steps 100
for k in 256 512 1024 2048 4096 1 7936; do run --steps 100 --opt dbg_skip=$k; done
run --steps 100 --opt weight_prefetch=0
for k in 256 512 1024 2048 4096 1 7936; do run --steps 100 --opt weight_prefetch=0 --opt dbg_skip=$k; done
} > $O/ablate3.txt 2>&1
cat $O/ablate3.txt
