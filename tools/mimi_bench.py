#!/usr/bin/env python
"""Mimi decode timing (row f-2): kyutai/mimi architecture, seeded synthetic weights, codes -> 24 kHz waveform on the device.
Every line carries its roofline: one-shot decodes against the fp32 matrix-pipe peak (the GEMMs run on the exact-fp32 MFMA,
157 TFLOP/s dense on MI355X; flops from csm_hf_amd.mimi.decode_gemm_work = the GEMM list of csrc/mimi.hip), the one-frame
streaming call against HBM (its 161 MB of fp32 weights are read once per call; 8 TB/s).
usage: python tools/mimi_bench.py [frames ...]   (run under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from csm_hf_amd import MimiDecoder, MimiDecodeConfig  # noqa: E402
from csm_hf_amd.mimi import synth_mimi_state_dict, decode_gemm_work  # noqa: E402

FP32_MFMA_PEAK_TF, HBM_PEAK_GBS = 157.0, 8000.0

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
    fl, wb = decode_gemm_work(cfg, T)
    tf = fl / min(ts) / 1e12
    print(f"{T:4d} frames = {sec:6.2f} s of audio ({out.shape[-1]} samples): {min(ts) * 1e3:7.2f} ms  = {sec / min(ts):7.0f} x real time"
          f"   | {fl / 1e9:7.1f} GFLOP -> {tf:6.2f} TFLOP/s = {tf / FP32_MFMA_PEAK_TF:.3f} of the fp32 matrix peak", flush=True)
    print(json.dumps({"workload": f"mimi decode, kyutai/mimi shape, {T} frames one shot, B=1", "ms": round(min(ts) * 1e3, 3),
                      "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                   "frac": round(tf / FP32_MFMA_PEAK_TF, 4), "flops": fl}}), flush=True)
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
med = ts[len(ts) // 2]
fl1, wb1 = decode_gemm_work(cfg, 1)
print(f"streaming, 1 frame (80 ms of audio) per call: median {med * 1e3:.2f} ms per call = {0.08 / med:.0f} x real time"
      f"   | {wb1 / 1e6:.0f} MB of weights -> {wb1 / med / 1e9:.0f} GB/s = {wb1 / med / 1e9 / HBM_PEAK_GBS:.3f} of 8 TB/s", flush=True)
print(json.dumps({"workload": "mimi stream_decode, kyutai/mimi shape, 1 frame per call", "ms": round(med * 1e3, 3),
                  "roofline": {"bound": "hbm", "achieved": round(wb1 / med / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(wb1 / med / 1e9 / HBM_PEAK_GBS, 4), "weight_bytes": wb1}}), flush=True)
# stream groups (csm_mimi_streams_*): S streams, one frame each per call, every launch covering all of them
for S in (1, 4, 16, 64):
    if S > dec.max_frames:
        break
    gc = torch.randint(0, cfg.codebook_size, (S, cfg.num_quantizers, 40), generator=g).to("cuda:0")
    dec.streams_open(S)
    ts = []
    for t in range(40):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dec.streams_decode(gc[:, :, t:t + 1])
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[8:])
    mg = ts[len(ts) // 2]
    print(f"stream group, {S:3d} streams x 1 frame per call: median {mg * 1e3:.2f} ms per call = {mg * 1e3 / S:.3f} ms per stream "
          f"({S * med / mg:.1f} x the throughput of {S} single-stream calls); {0.08 / mg * S:.0f} stream-seconds per second", flush=True)
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
