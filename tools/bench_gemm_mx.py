#!/usr/bin/env python
"""MX-fp8 prefill GEMM (csrc/gemm_mx.h, v_mfma_scale_f32_16x16x128_f8f6f4) through the C ABI (csm_gemm_mx) at the backbone's
shapes: us per launch, TFLOP/s, fraction of the dense MX-fp8 matrix peak (5 PFLOP/s nominal; 4.66 PFLOP/s micro-benchmark
ceiling in the CDNA guide), next to the bf16 GEMM of the same shape (csm_gemm, fp32 activations -> exact three-plane path).
usage: python tools/bench_gemm_mx.py [R ...] [option=value ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from csm_hf_amd import CSMConfig
from csm_hf_amd.engine import Engine, _ptr, _ck, quantize_mx_rows
from csm_hf_amd.synth import synth_state_dict

PEAK = 5000.0
cfg = CSMConfig.tiny()
eng = Engine(cfg, synth_state_dict(cfg), "cuda:0", torch.float32, max_batch=1, max_len=64, max_frames=4, max_prefill_rows=128)
shapes = [("bb qkv", 3072, 2048), ("bb o", 2048, 2048), ("bb gate/up", 16384, 2048), ("bb down", 2048, 8192), ("square", 4096, 4096)]
Rs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [512, 2048, 4096]
for k, v in (a.split("=") for a in sys.argv[1:] if "=" in a):      # engine options, e.g. gemm_256=1 (the 256 x 256 tile)
    eng.set_option(k, int(v))
print("| shape | N | K | " + " | ".join(f"R={r}: us (TFLOP/s, frac of 5 PF)" for r in Rs) + " |")
print("|---|---|---|" + "---|" * len(Rs))
for name, N, K in shapes:
    Wq, Ws = quantize_mx_rows(torch.randn(N, K, device="cuda") * 0.05)
    cells = []
    for R in Rs:
        Aq, As = quantize_mx_rows(torch.randn(R, K, device="cuda"))
        out = torch.empty(R, N, device="cuda")
        torch.cuda.synchronize()

        def run():
            _ck(eng.lib, eng.lib.csm_gemm_mx(eng._h, _ptr(Wq), _ptr(Ws), N, K, _ptr(Aq), _ptr(As), R, _ptr(out)))
        for _ in range(3):
            run()
        eng.sync()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            run()
        eng.sync()
        us = (time.perf_counter() - t0) / n * 1e6
        tf = 2.0 * R * N * K / us / 1e6
        cells.append(f"{us:.0f} ({tf:.0f}, {tf / PEAK:.3f})")
    print(f"| {name} | {N} | {K} | " + " | ".join(cells) + " |", flush=True)
