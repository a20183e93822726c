# A/B: 64 / 32-row workgroups of the LDS-DMA GEMM (gemm_dma_skinny) for the split-K / SwiGLU launches of short prefills
for ctx in 16 32 64 128 192 256 384 512; do for mode in 0 1; do for v in 0 1; do
  timeout 200 python tools/prefill_bench.py $ctx 1 9 $mode gemm_dma_skinny=$v 2>&1 | grep "^ctx"
done; done; done > gpurun_out/skinny.txt
cat gpurun_out/skinny.txt
