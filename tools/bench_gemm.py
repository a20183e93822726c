#!/usr/bin/env python
"""Prefill GEMM timing through the C ABI (csm_gemm): C[R,N] = A[R,K] @ W[N,K]^T, fp32 activations, bf16 weights.
usage: python tools/bench_gemm.py [R ...]   -> TFLOP/s useful and on the matrix pipe (x3 for the exact split)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from csm_hf_amd import CSMConfig
from csm_hf_amd.engine import Engine, _ptr, _ck, DT_BF16
from csm_hf_amd.synth import synth_state_dict

cfg = CSMConfig.tiny()
eng = Engine(cfg, synth_state_dict(cfg), "cuda:0", torch.float32, max_batch=1, max_len=64, max_frames=4, max_prefill_rows=128)
shapes = [("bb qkv", 3072, 2048), ("bb o", 2048, 2048), ("bb gate/up", 16384, 2048), ("bb down", 2048, 8192)]
Rs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [512, 2048, 8192]
print("| shape | N | K | " + " | ".join(f"R={r}: us (TF/s useful / MFMA)" for r in Rs) + " |")
print("|---|---|---|" + "---|" * len(Rs))
for name, N, K in shapes:
    W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    cells = []
    for R in Rs:
        A = torch.randn(R, K, device="cuda")
        out = torch.empty(R, N, device="cuda")
        torch.cuda.synchronize()
        def run():
            _ck(eng.lib, eng.lib.csm_gemm(eng._h, _ptr(W), DT_BF16, None, N, K, _ptr(A), R, _ptr(out)))
        for _ in range(3):
            run()
        eng.sync()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            run()
        eng.sync()
        us = (time.perf_counter() - t0) / n * 1e6
        tf = 2.0 * R * N * K / us / 1e6
        cells.append(f"{us:.0f} ({tf:.0f} / {3 * tf:.0f})")
    print(f"| {name} | {N} | {K} | " + " | ".join(cells) + " |")
