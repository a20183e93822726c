#!/usr/bin/env python
"""Real-time streaming, end to end on the device: every `generate_frame` call (csm-1b, B = 1, reference loop of
modeling_csm.py:644-690) is followed by the streaming Mimi decode of that frame -> 80 ms of 24 kHz audio per step.
Synthetic weights (no checkpoints in the image); prints per-step latency against the 80 ms real-time budget.
usage: python tools/stream_demo.py [frames] [context_frames]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from csm_hf_amd import CSMConfig, CSMModel, MimiDecoder, MimiDecodeConfig  # noqa: E402
from csm_hf_amd.mimi import synth_mimi_state_dict  # noqa: E402
from csm_hf_amd.synth import synth_state_dict, synth_context  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = "cuda:0"
cfg = CSMConfig()
m = CSMModel(cfg)
m.load_state_dict(synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev, bf16_representable=True))
m = m.eval()
mc = MimiDecodeConfig()
dec = MimiDecoder(mc, synth_mimi_state_dict(mc, seed=0, device=dev), dev, max_frames=8)
ids, mask = synth_context(cfg, 1, ctx // 4, ctx - ctx // 4, seed=2)
cur, cm, pkv = ids.to(dev), mask.to(dev), None
m.setup_caches(1)
dec.stream_reset()
t_gen, t_wav, samples = [], [], 0
for i in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m.generate_frame(cur, cm, temperature=0.9, topk=50, past_key_values=pkv, return_dict=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    codes = out.samples.clamp(max=mc.codebook_size - 1).t().contiguous()      # [32, 1]; synthetic tokens may exceed the codec's 2048 entries
    wav = dec.stream_decode(codes)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    t_gen.append(t1 - t0)
    t_wav.append(t2 - t1)
    samples += wav.shape[-1]
    pkv = out.past_key_values
    cur = torch.cat([out.samples, torch.zeros(1, 1, dtype=torch.long, device=dev)], 1).unsqueeze(1)
    cm = torch.zeros(1, 1, 33, dtype=mask.dtype, device=dev)
    cm[:, :, :32] = 1
med = lambda v: sorted(v)[len(v) // 2] * 1e3
steady_g, steady_w = t_gen[5:], t_wav[5:]
print(f"{n} frames after a {ctx}-frame context: first step (prefill + frame + waveform) {(t_gen[0] + t_wav[0]) * 1e3:.1f} ms; then per 80 ms frame: "
      f"generate_frame median {med(steady_g):.2f} ms + waveform {med(steady_w):.2f} ms = {med(steady_g) + med(steady_w):.2f} ms "
      f"(max {max(a + b for a, b in zip(steady_g, steady_w)) * 1e3:.2f} ms) = {80.0 / (med(steady_g) + med(steady_w)):.1f} x real time; {samples} samples")

# ---- a BATCH streamed to audio frame by frame: B rows through the same generate_frame loop, every frame of every row decoded
# by ONE stream-group call (csm_mimi_streams_*: all B streams in every launch) ----
for B in (16, 64):
    gdec = MimiDecoder(mc, None, dev, max_frames=max(8, B), _packed=dec.packed)      # the same device weights
    gdec.streams_open(B)
    ids, mask = synth_context(cfg, B, ctx // 4, ctx - ctx // 4, seed=3)
    cur, cm, pkv = ids.to(dev), mask.to(dev), None
    m.setup_caches(B)
    t_gen, t_wav = [], []
    nb = min(n, 40)
    for i in range(nb):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = m.generate_frame(cur, cm, temperature=0.9, topk=50, past_key_values=pkv, return_dict=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        codes = out.samples.clamp(max=mc.codebook_size - 1).unsqueeze(-1).contiguous()       # [B, 32, 1]
        wav = gdec.streams_decode(codes)                                                     # [B, 1, 1920]
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        t_gen.append(t1 - t0)
        t_wav.append(t2 - t1)
        pkv = out.past_key_values
        cur = torch.cat([out.samples, torch.zeros(B, 1, dtype=torch.long, device=dev)], 1).unsqueeze(1)
        cm = torch.zeros(B, 1, 33, dtype=mask.dtype, device=dev)
        cm[:, :, :32] = 1
    g, w = med(t_gen[5:]), med(t_wav[5:])
    print(f"batch of {B} rows, {nb} frames each: generate_frame median {g:.2f} ms + stream-group waveform {w:.2f} ms = {g + w:.2f} ms per "
          f"80 ms frame-step = {80.0 / (g + w):.1f} x real time for every one of the {B} streams ({B * 1e3 / (g + w):.0f} frames/s with audio; "
          f"{B} single-stream codec calls would add {B * med(steady_w):.1f} ms)")
    gdec.close()
