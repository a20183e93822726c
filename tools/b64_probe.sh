# batches of 33-64 rows: tests + frame-step with rows64 on / off -> profiles/r03_b64_rows64.txt (run through gpurun)
O=gpurun_out/b64; mkdir -p $O
run() { timeout 300 python bench.py --no-cpu-baseline --config4 0 --lean "$@" 2>>$O/err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-60s ms/step %.4f  frames/s %.1f' % (' '.join(sys.argv[1:]), d['ms_per_step'], d['value']))
" "$@"; }
{
timeout 900 python -m pytest tests/test_gpu_generate.py -x -q -k "64_row or other_shapes" 2>&1 | tail -5
run --batch 48 --steps 50
run --batch 48 --steps 50 --opt rows64=0
run --batch 64 --steps 50
run --batch 64 --steps 50 --opt rows64=0
} > $O/b64b.txt 2>&1
cat $O/b64b.txt; tail -3 $O/err.log
