# A/B of the fused prefill epilogues (GEPI_ROPE in the QKV GEMM, MX output of the flash attention): whole prefills, option off / on
O=gpurun_out; mkdir -p $O
{
for ctx in 64 512 2048; do
  for mode in 1 2; do
    timeout 200 python tools/prefill_bench.py $ctx 1 9 $mode prefill_fuse_rope=0 prefill_fuse_quant=0
    timeout 200 python tools/prefill_bench.py $ctx 1 9 $mode prefill_fuse_rope=1 prefill_fuse_quant=0
    [ $mode = 2 ] && timeout 200 python tools/prefill_bench.py $ctx 1 9 $mode prefill_fuse_rope=1 prefill_fuse_quant=1
  done
done
timeout 200 python tools/prefill_bench.py 2048 1 5 0 prefill_fuse_rope=0
timeout 200 python tools/prefill_bench.py 2048 1 5 0 prefill_fuse_rope=1
timeout 300 python tools/prefill_bench.py 512 16 5 1 prefill_fuse_rope=0
timeout 300 python tools/prefill_bench.py 512 16 5 1 prefill_fuse_rope=1
timeout 300 python tools/prefill_bench.py 512 16 5 2 prefill_fuse_rope=0 prefill_fuse_quant=0
timeout 300 python tools/prefill_bench.py 512 16 5 2 prefill_fuse_rope=1 prefill_fuse_quant=1
} 2>&1 | grep -v amdgpu.ids > $O/fuse_probe.txt
cat $O/fuse_probe.txt
