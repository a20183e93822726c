# end-of-round check on one MI355X box: the whole GPU suite, the driver's bench line, config 5, the prefill table, the serving bench
O=gpurun_out/final; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/gputest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
for c in 32 64 128 256 512 1024 2048; do for m in 0 1 2; do timeout 200 python tools/prefill_bench.py $c 1 8 $m; done; done 2>&1 | grep "^ctx" > $O/prefill.txt
for m in 1 2; do timeout 300 python tools/prefill_bench.py 512 16 3 $m 2>&1 | grep "^ctx" >> $O/prefill.txt; done
timeout 300 python bench.py --no-cpu-baseline --config4 0 --weights fp8 --ctx 2048 --steps 500 --warmup 4 > $O/bench_config5.json 2>> $O/bench.err
{ timeout 400 python tools/serve_bench.py 64 16 2>&1 | grep -v amdgpu | tail -4; timeout 400 python tools/serve_bench.py 256 64 2>&1 | grep -v amdgpu | tail -4; } > $O/serve.txt
tail -3 $O/gputest.log; cat $O/prefill.txt $O/serve.txt; cut -c1-300 $O/bench.json
