# A/B + knock-out timing of the decode chain (results of dbg_skip runs are WRONG by construction: timing only)
O=gpurun_out/ablate; mkdir -p $O
run() { timeout 200 python bench.py --no-cpu-baseline --config4 0 --lean "$@" 2>>$O/err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-60s ms/step %.4f  frames/s %.1f  checksum %s' % (' '.join(sys.argv[1:]), d['ms_per_step'], d['value'], d.get('tokens_checksum_per_rank')))
" "$@"; }
{
python -m pytest tests/test_gpu_round3.py -x -q -k "torchs_global" 2>&1 | tail -3
run --batch 16 --steps 100
run --batch 16 --steps 100 --opt attn_prefetch=2
run --batch 16 --steps 100 --opt nsplit_backbone=16
run --batch 16 --steps 100 --opt nsplit_backbone=32
run --batch 16 --steps 100 --opt nsplit_backbone=4 --opt attn_prefetch=2
run --batch 16 --steps 100 --opt nsplit_backbone=16 --opt attn_prefetch=2
run --batch 16 --steps 100 --kv-dtype bf16
run --batch 16 --steps 100 --kv-dtype bf16 --opt attn_prefetch=2
run --batch 16 --steps 100 --kv-dtype bf16 --opt nsplit_backbone=16
run --batch 16 --steps 100 --kv-dtype bf16 --opt nsplit_backbone=4 --opt attn_prefetch=2
run --steps 100
run --steps 100 --opt attn_prefetch=1
run --steps 100 --opt nsplit_backbone=32
run --steps 100 --opt nsplit_backbone=8 --opt attn_prefetch=1
run --steps 100 --kv-dtype bf16
} > $O/ablate2.txt 2>&1
cat $O/ablate2.txt
