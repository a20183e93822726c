#!/usr/bin/env python
"""Time of one training step's forward + backward (csm_forward_backward, row f-3) on csm-1b, bf16 checkpoint, fp32 arithmetic:
python tools/train_bench.py [B] [S] [reps].  Context: S/8 text frames + audio frames, every audio frame fully labelled (the
reference's processor amortises the decoder loss over 1/16 of the frames; `--amortize N` labels every N-th frame only).
Roofline note: the pass is built for correctness (csrc/train.h): its matrix products run on the exact fp32 / three-plane
paths and the attention kernels are one wavefront per (row, head)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from csm_hf_amd import CSMConfig, CSMModel  # noqa: E402
from csm_hf_amd.synth import synth_state_dict, synth_context  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if len(args) > 0 else 2
S = int(args[1]) if len(args) > 1 else 128
reps = int(args[2]) if len(args) > 2 else 3
amort = int(sys.argv[sys.argv.index("--amortize") + 1]) if "--amortize" in sys.argv else 1
dev = torch.device("cuda:0")
cfg = CSMConfig()
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev, bf16_representable=True)
m = CSMModel(cfg)
m.load_state_dict(sd)
del sd
nt = S // 8
ids, mask = synth_context(cfg, B, nt, S - nt, seed=3)
labels = torch.full_like(ids, -100)
labels[:, nt:, :32] = ids[:, nt:, :32]
if amort > 1:
    for t in range(nt, S):
        if (t - nt) % amort:
            labels[:, t, 1:32] = -100
ids, mask, labels = ids.to(dev), mask.to(dev), labels.to(dev)
ts = []
for _ in range(reps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, grads = m.loss_and_grads(ids, mask, labels)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
    del grads
nf = int((labels[:, :, :32] != -100).all(-1).sum())
# 6 flops per parameter per token (2 forward + 4 backward) over the backbone rows and the 32-position decoder rows
flops = 6.0 * (973e6 + 8.4e6 / 2) * B * S + 6.0 * 111e6 * nf * 32
print(f"csm-1b B={B} S={S} labelled decoder frames {nf}: loss {float(out.loss):.4f}; forward+backward min {min(ts[1:]) * 1e3:.1f} ms "
      f"(first call {ts[0] * 1e3:.0f} ms incl. transposed weight copies) = {flops / min(ts[1:]) / 1e12:.1f} TFLOP/s of model flops", flush=True)
