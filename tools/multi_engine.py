#!/usr/bin/env python
"""Experiment: N independent engines of B rows each (own stream + hipGraph, SHARED weights) running concurrently on one GPU.
A frame-step is a chain of ~650-800 latency-bound launches that leaves most of the chip idle, so independent
chains on separate streams can overlap.  usage: python tools/multi_engine.py [--batch B] [n_engines ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from csm_hf_amd import CSMConfig
from csm_hf_amd.engine import Engine
from csm_hf_amd.synth import synth_state_dict, synth_context

B = 1
if "--batch" in sys.argv:
    i = sys.argv.index("--batch"); B = int(sys.argv[i + 1]); del sys.argv[i:i + 2]
cfg = CSMConfig()
dev = torch.device("cuda:0")
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev, bf16_representable=True)
K, W = 200, 8
first = Engine(cfg, sd, dev, torch.bfloat16, max_batch=B, max_len=512 + K + W + 2, max_frames=K + W + 1, max_prefill_rows=512 * B)
del sd
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    engs = [first] + [Engine(cfg, None, dev, torch.bfloat16, max_batch=B, max_len=512 + K + W + 2, max_frames=K + W + 1,
                             max_prefill_rows=512 * B, packed=first.packed) for _ in range(n - 1)]
    ids, mask = synth_context(cfg, n * B, 128, 384, seed=2)
    samp = []
    for i, e in enumerate(engs):
        e.reset(); e.set_kv_start([0] * B); e.prefill(ids[i * B:(i + 1) * B], mask[i * B:(i + 1) * B], want_outputs=False)
        s = e.sampling(temperature=1.0, topk=1); samp.append(s)
        e.generate(s, W, True); e.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e, s in zip(engs, samp):
        e.generate(s, K, True)
    for e in engs:
        e.sync()
    dt = time.perf_counter() - t0
    toks = [e.read_frames(0, W + K).cpu() for e in engs]
    print(f"{n} engines x {B} rows: {K} frames each in {dt*1e3:.1f} ms -> {n*B*K/dt:.1f} frames/s aggregate, {dt/K*1e3:.3f} ms per frame-step round", flush=True)
    for e in engs[1:]:
        e.close()
