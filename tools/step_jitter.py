#!/usr/bin/env python
"""B = 1: distribution of the frame-step time, one graph replay at a time (HIP events inside csm_generate): are there stalls inside
the chain outside the profiler?  usage: python tools/step_jitter.py [steps] [chunk]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from csm_hf_amd import CSMConfig, CSMModel
from csm_hf_amd.synth import synth_state_dict, synth_context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
cfg = CSMConfig()
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev, bf16_representable=True)
m = CSMModel(cfg); m.load_state_dict(sd); del sd
ids, mask = synth_context(cfg, 1, 128, 384, seed=1)
eng = m._ensure_engine(1, 512 + n * chunk + 40, n * chunk + 32, 512)
eng.reset(); eng.set_kv_start([0])
eng.prefill(ids, mask)
s = eng.sampling(temperature=1.0, topk=1, seed=7)
eng.generate(s, 16, True)     # warm-up (graph capture)
ts = []
for _ in range(n):
    eng.generate(s, chunk, True)
    ts.append(eng.last_generate_ms() / chunk)
ts = sorted(ts)
q = lambda p: ts[min(len(ts) - 1, int(p * len(ts)))]
print(f"{n} x {chunk}-frame replays: min {ts[0]:.4f}  p10 {q(0.1):.4f}  median {q(0.5):.4f}  p90 {q(0.9):.4f}  p99 {q(0.99):.4f}  max {ts[-1]:.4f}  mean {sum(ts)/len(ts):.4f} ms per step")
