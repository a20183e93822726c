#!/usr/bin/env python
"""Mimi decode timing (row f-2): kyutai/mimi architecture, seeded synthetic weights, codes -> 24 kHz waveform on the device.
usage: python tools/mimi_bench.py [frames ...]   (run under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from csm_hf_amd import MimiDecoder, MimiDecodeConfig  # noqa: E402
from csm_hf_amd.mimi import synth_mimi_state_dict  # noqa: E402

frames = [int(a) for a in sys.argv[1:]] or [25, 100, 200, 500]
cfg = MimiDecodeConfig()
sd = synth_mimi_state_dict(cfg, seed=0, device="cuda:0")
dec = MimiDecoder(cfg, sd, "cuda:0", max_frames=max(frames))
g = torch.Generator().manual_seed(1)
for T in frames:
    codes = torch.randint(0, cfg.codebook_size, (1, cfg.num_quantizers, T), generator=g).to("cuda:0")
    dec.decode(codes)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = dec.decode(codes)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    sec = T / 12.5
    print(f"{T:4d} frames = {sec:6.2f} s of audio ({out.shape[-1]} samples): {min(ts) * 1e3:7.2f} ms  = {sec / min(ts):7.0f} x real time", flush=True)
# streaming: one frame per call, the state a real-time stream carries (what follows every generate_frame)
dec.stream_reset()
codes = torch.randint(0, cfg.codebook_size, (cfg.num_quantizers, 64), generator=g).to("cuda:0")
ts = []
for t in range(64):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dec.stream_decode(codes[:, t:t + 1])
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
ts = sorted(ts[8:])
print(f"streaming, 1 frame (80 ms of audio) per call: median {ts[len(ts) // 2] * 1e3:.2f} ms per call = {0.08 / ts[len(ts) // 2]:.0f} x real time", flush=True)
try:
    from transformers import MimiConfig, MimiModel
    m = MimiModel(MimiConfig()).eval()
    codes = torch.randint(0, cfg.codebook_size, (1, cfg.num_quantizers, 100), generator=g)
    with torch.no_grad():
        m.decode(codes)
        t0 = time.perf_counter()
        m.decode(codes)
        dt = time.perf_counter() - t0
    print(f"transformers MimiModel.decode on the host CPU ({torch.get_num_threads()} threads), 100 frames: {dt * 1e3:.0f} ms = {8.0 / dt:.1f} x real time")
except Exception as ex:   # transformers without Mimi: the GPU numbers stand alone
    print("transformers MimiModel not available:", ex)
