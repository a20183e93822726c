# kernel traces of the B = 1 and B = 16 bench + tools/step_timeline.py --gaps: which launch boundaries show gaps under the profiler (run through gpurun)
O=gpurun_out/gaps; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
timeout 300 python bench.py --no-cpu-baseline --config4 0 > $O/b1.json 2> $O/b1.err
timeout 300 python bench.py --no-cpu-baseline --config4 0 --batch 16 --steps 100 > $O/b16.json 2>> $O/b1.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/$O/s1 -o t -- python $R/bench.py --steps 20 --warmup 4 --lean --opt weight_prefetch=0 > $R/$O/s1.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d $R/$O/s16 -o t -- python $R/bench.py --batch 16 --steps 20 --warmup 4 --lean > $R/$O/s16.log 2>&1
cd $R
python tools/step_timeline.py $O/s1/t_results.db --gaps 3 > $O/b1_gaps.md 2>&1
python tools/step_timeline.py $O/s16/t_results.db --gaps 3 > $O/b16_gaps.md 2>&1
rm -rf $O/s1 $O/s16
cat $O/b1.json $O/b16.json | cut -c1-400
