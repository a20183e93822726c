# gemm256_kernel (csrc/gemm256.h): parity test + MX GEMM microbenchmark + whole-prefill timings by schedule variant -> profiles/r03_gemm256.txt (run through gpurun)
O=gpurun_out/g256; mkdir -p $O
{
timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -k "gemm256" 2>&1 | tail -5
for o in 16777217 1; do echo "gemm_256=$o"; timeout 300 python tools/bench_gemm_mx.py 2048 4096 8192 gemm_256=$o 2>&1 | grep "^|"; done
for o in 16777472 256; do
  timeout 200 python tools/prefill_bench.py 2048 1 8 1 gemm_256=$o 2>&1 | grep "^ctx"
  timeout 200 python tools/prefill_bench.py 2048 1 8 2 gemm_256=$o 2>&1 | grep "^ctx"
  timeout 300 python tools/prefill_bench.py 512 16 4 1 gemm_256=$o 2>&1 | grep "^ctx"
  timeout 300 python tools/prefill_bench.py 512 16 4 2 gemm_256=$o 2>&1 | grep "^ctx"
done
} > $O/ubench3.txt 2>&1
cat $O/ubench3.txt
