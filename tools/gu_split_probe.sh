# A/B: gate/up GEMM of a short prefill split over K (prefill_splitk_gu) + swiglu_reduce_kernel
for ctx in 32 64 128; do for mode in 0 1 2; do for v in 0 2 4; do
  timeout 200 python tools/prefill_bench.py $ctx 1 9 $mode prefill_splitk_gu=$v 2>&1 | grep "^ctx"
done; done; done > gpurun_out/gu_split.txt
cat gpurun_out/gu_split.txt
