run() { timeout 300 python bench.py --no-cpu-baseline --config4 0 --lean "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-50s ms/step %.4f  frames/s %.1f' % (' '.join(sys.argv[1:]), d['ms_per_step'], d['value']))
" "$@"; }
mkdir -p gpurun_out/b64
{
python -m pytest tests/test_gpu_generate.py -x -q -k "64_row or other_shapes" 2>&1 | tail -2
run --batch 32 --steps 50
run --batch 32 --steps 50 --opt nsplit_backbone=8
run --batch 32 --steps 50 --opt nsplit_backbone=4
run --batch 48 --steps 50
run --batch 64 --steps 50
run --batch 64 --steps 50 --ctx 2048
run --batch 64 --steps 50 --ctx 2048 --opt nsplit_backbone=8
} > gpurun_out/b64/b64d.txt 2>&1
cat gpurun_out/b64/b64d.txt
