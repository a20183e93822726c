mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_round2.py -q -k "streamer or graph" 2>&1 | tail -15 > gpurun_out/t_pf.txt
for o in "weight_prefetch=0" "weight_prefetch=1" "prefetch_window_mb=16" "prefetch_window_mb=28" "prefetch_sub_kb=2048" "prefetch_sub_kb=8192" "prefetch_grid=512"; do
  echo "== $o" >> gpurun_out/pf_bench.txt
  timeout 300 python bench.py --steps 100 --warmup 10 --lean --opt $o 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['hip_event_ms_per_step'], d['roofline']['frac'], d.get('parity'))
" >> gpurun_out/pf_bench.txt
done
cat gpurun_out/t_pf.txt gpurun_out/pf_bench.txt
