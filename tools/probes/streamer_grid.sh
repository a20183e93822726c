cfgs=""
for sub in 4096 6144 8192 10240 12288; do for win in 4 6 8 12; do for st in 0 64; do cfgs="$cfgs prefetch_sub_kb=$sub,prefetch_window_mb=$win,prefetch_stride=$st"; done; done; done
python tools/ab_bench.py --steps 80 $cfgs 2>/dev/null
