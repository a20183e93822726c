# gemm256_kernel (csrc/gemm256.h): MX GEMM microbenchmark by schedule variant / knock-out -> profiles/r03_gemm256.txt (run through gpurun)
O=gpurun_out/g256; mkdir -p $O
{
for o in 1 50331649 67108865 83886081; do echo "gemm_256=$o"; timeout 300 python tools/bench_gemm_mx.py 4096 8192 gemm_256=$o 2>&1 | grep "^|" | grep -v "^| shape\|^|---"; done
} > $O/knock.txt 2>&1
cat $O/knock.txt
