O=gpurun_out/b16; R=$PWD; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/stats -o r01 -- python $R/bench.py --steps 10 --warmup 2 --lean --batch 16 > $R/$O/stats.log 2>&1
cd $R
python tools/rocprof_summary.py $O/stats/r01_results.db 12 > $O/kernel_stats.md 2>&1
rm -rf $O/stats
grep "gemm16\|attn_decode\|attn_combine\|sample\|embed" $O/kernel_stats.md | head -60
