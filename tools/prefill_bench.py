#!/usr/bin/env python
"""Prefill timing of csm-1b (bf16 weights) in both precisions: python tools/prefill_bench.py [ctx] [batch] [reps] [mode 0|1|2]
(run under `rocprofv3 --kernel-trace --stats` for the per-kernel split)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from csm_hf_amd import CSMConfig, CSMModel  # noqa: E402
from csm_hf_amd.synth import synth_state_dict, synth_context  # noqa: E402

ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
modes = [int(sys.argv[4])] if len(sys.argv) > 4 else [0, 1]
opts = [a.split("=") for a in sys.argv[5:] if "=" in a]       # engine options name=value (A/B)
dev = torch.device("cuda:0")
cfg = CSMConfig()
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev, bf16_representable=True)
m = CSMModel(cfg)
m.load_state_dict(sd)
del sd
ids, mask = synth_context(cfg, B, ctx // 4, ctx - ctx // 4, seed=2)
eng = m._ensure_engine(B, ctx + 8, 4, B * ctx)
for k, v in opts:
    eng.set_option(k, int(v))
for mode in modes:      # 0 exact, 1 bf16 activations, 2 MX-fp8 weights and activations (gemm_mx.h)
    eng.set_option("prefill_bf16", 1 if mode else 0)
    if mode == 2 and not eng.has_mx:
        eng.enable_mx(m.state_dict())
    eng.set_option("prefill_mx", 1 if mode == 2 else 0)
    ts = []
    for _ in range(reps):
        eng.reset()
        eng.set_kv_start([0] * B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.prefill(ids, mask, want_outputs=False)
        eng.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    flops = 2 * 973e6 * B * ctx
    name = ("exact", "bf16", "mxfp8")[mode]
    ostr = " ".join("=".join(o) for o in opts)
    print(f"ctx {ctx} B {B} mode={name} {ostr}: min {min(ts):.2f} ms  median {sorted(ts)[len(ts) // 2]:.2f} ms  "
          f"({flops / min(ts) / 1e9:.0f} TFLOP/s on the GEMM flops alone)", flush=True)
