# context attention with one / two key groups per workgroup -> profiles/r03_attn_key_groups.txt (run through gpurun)
O=gpurun_out/attn; mkdir -p $O
{
timeout 900 python -m pytest tests/test_gpu_generate.py tests/test_gpu_round2.py tests/test_gpu_round3.py -x -q -k "prefill_precision or bf16 or mxfp8 or gemm256 or lds_dma" 2>&1 | tail -5
for o in 1 0 3; do
  timeout 200 python tools/prefill_bench.py 512 1 8 1 attn_key_groups=$o 2>&1 | grep "^ctx"
  timeout 200 python tools/prefill_bench.py 2048 1 8 1 attn_key_groups=$o 2>&1 | grep "^ctx"
  timeout 200 python tools/prefill_bench.py 2048 1 8 2 attn_key_groups=$o 2>&1 | grep "^ctx"
  timeout 300 python tools/prefill_bench.py 512 16 4 1 attn_key_groups=$o 2>&1 | grep "^ctx"
done
} > $O/attn.txt 2>&1
cat $O/attn.txt
