"""exact-mode prefill on gemm_dma3_k32_kernel (32-wide k-steps, two workgroups per CU) against the 64-wide three-plane LDS-DMA
kernel: bitwise equality without a K split, then timings.  usage: python tools/k32_probe.py   (csm-1b, one MI355X)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csm_hf_amd import CSMConfig, CSMModel
from csm_hf_amd.synth import synth_state_dict, synth_context
dev = torch.device("cuda:0")
cfg = CSMConfig()
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev, bf16_representable=True)
m = CSMModel(cfg); m.load_state_dict(sd); del sd
ids, mask = synth_context(cfg, 1, 128, 384, seed=2)
def run(opts):
    eng = m._ensure_engine(1, 600, 4, 512)
    for k, v in opts.items(): eng.set_option(k, v)
    eng.reset(); eng.set_kv_start([0])
    lh, lg = eng.prefill(ids, mask)
    return lh.cpu(), lg.cpu()
a = run(dict(gemm_dma=8, prefill_splitk=0))
for rep in range(3):
    b = run(dict(gemm_dma=40, prefill_splitk=0))
    print("k32 bitwise == 64-wide three-plane kernel:", torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), float((a[0]-b[0]).abs().max()))
c = run(dict(gemm_dma=0, prefill_splitk=0))
print("vs square tile:", torch.equal(a[0], c[0]), float((a[0]-c[0]).abs().max()))
m._engine.set_option("gemm_dma", 5); m._engine.set_option("prefill_splitk", 1)
