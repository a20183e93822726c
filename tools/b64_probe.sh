run() { timeout 300 python bench.py --no-cpu-baseline --config4 0 --lean "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-50s ms/step %.4f  frames/s %.1f' % (' '.join(sys.argv[1:]), d['ms_per_step'], d['value']))
" "$@"; }
mkdir -p gpurun_out/b64
{
for n in 8 4 2 8 4 2; do run --batch 16 --steps 100 --opt nsplit_backbone=$n; done
for n in 8 4 2; do run --batch 16 --steps 100 --ctx 2048 --opt nsplit_backbone=$n; done
} > gpurun_out/b64/b16split.txt 2>&1
cat gpurun_out/b64/b16split.txt
