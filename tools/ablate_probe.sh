# A/B + knock-out timing of the decode chain (results of dbg_skip runs are WRONG by construction: timing only)
O=gpurun_out/ablate; mkdir -p $O
run() { timeout 200 python bench.py --no-cpu-baseline --config4 0 --lean "$@" 2>>$O/err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-60s ms/step %.4f  frames/s %.1f  checksum %s parity %s' % (' '.join(sys.argv[1:]), d['ms_per_step'], d['value'], d.get('tokens_checksum_per_rank'), (d.get('parity') or {}).get('equal_all')))
" "$@"; }
{
for v in 0 1 2 3 0 1 3; do run --steps 200 --opt gemv_norm_ks=$v; done
for v in 0 3; do run --steps 200 --opt weight_prefetch=0 --opt gemv_norm_ks=$v; done
for v in 0 3; do run --steps 200 --ctx 2048 --opt gemv_norm_ks=$v; done
} > $O/normks.txt 2>&1
cat $O/normks.txt
