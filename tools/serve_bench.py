#!/usr/bin/env python
"""Continuous batching (csm_hf_amd.serving.ContinuousBatcher) against static batches on csm-1b: N utterances with
512-frame contexts... kept short here: contexts of 64-128 frames, frame budgets 20-200, batch of 16 rows.
usage: python tools/serve_bench.py [n_utterances] [batch] [audio]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from csm_hf_amd import CSMConfig, CSMModel, ContinuousBatcher  # noqa: E402
from csm_hf_amd.synth import synth_state_dict, synth_context  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
cfg = CSMConfig()
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev, bf16_representable=True)
m = CSMModel(cfg)
m.load_state_dict(sd)
del sd
g = torch.Generator().manual_seed(3)
reqs = []
for i in range(N):
    T = int(torch.randint(64, 129, (1,), generator=g))
    budget = int(torch.randint(20, 201, (1,), generator=g))
    ids, mask = synth_context(cfg, 1, T // 4, T - T // 4, seed=500 + i)
    reqs.append((ids[0], mask[0], budget))
total = sum(r[2] for r in reqs)
AUDIO = len(sys.argv) > 3 and sys.argv[3] == "audio"      # third argument "audio": every utterance also leaves as a waveform
dec = None
if AUDIO:
    from csm_hf_amd import MimiDecoder, MimiDecodeConfig
    from csm_hf_amd.mimi import synth_mimi_state_dict
    mc = MimiDecodeConfig()
    dec = MimiDecoder(mc, synth_mimi_state_dict(mc, seed=0, device=dev), dev, max_frames=max(64, 8 * B))   # one group call per chunk of 8 frames
for label in (("warm-up", "continuous", "continuous+audio") if AUDIO else ("warm-up", "continuous")):
    cb = ContinuousBatcher(m, batch_size=B, topk=1, check_every=8, audio_decoder=dec if label == "continuous+audio" else None,
                           clamp_audio_ids=True)     # synthetic weights emit ids 2048-2050 too
    for ids, mask, budget in (reqs[:B] if label == "warm-up" else reqs):
        cb.submit(ids, mask, max_new_frames=budget if label != "warm-up" else 16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = cb.run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if label != "warm-up":
        assert sum(v.shape[0] for v in out.values()) == total
        extra = ""
        if label == "continuous+audio":
            secs = sum(v.numel() for v in cb.audio.values()) / 24000.0
            extra = f"; {secs:.0f} s of 24 kHz audio decoded on the way = {secs / dt:.0f} x real time in total"
        print(f"{label:20s}: {N} utterances, {total} frames, batch {B}: {dt:.2f} s = {total / dt:.0f} useful frames/s "
              f"({cb.joined_mid_batch} utterances joined a running batch){extra}", flush=True)
        # latency against the real-time deadline of 80 ms per frame and stream (all utterances are submitted at t = 0: the
        # time to first frame of a queued utterance includes its wait for a free row)
        ls = cb.latency_summary()
        first = [cb.latency[r]["ttff_s"] for r in sorted(cb.latency)[:B]]
        ms = lambda v: "-" if v is None else f"{v * 1e3:.1f}"
        print(f"{'':20s}  time to first frame: first batch {ms(min(first))}-{ms(max(first))} ms; all: p50 {ms(ls['ttff_s']['p50'])} p99 {ms(ls['ttff_s']['p99'])} "
              f"max {ms(ls['ttff_s']['max'])} ms | inter-chunk gap (chunk = {cb.check_every} frames, deadline {ls['chunk_deadline_s'] * 1e3:.0f} ms): p50 "
              f"{ms(ls['inter_chunk_gap_s']['p50'])} p99 {ms(ls['inter_chunk_gap_s']['p99'])} max {ms(ls['inter_chunk_gap_s']['max'])} ms; late chunks "
              f"{ls['late_chunks']} of {ls['chunks']}; joins deferred by the per-chunk prefill budget {cb.joins_deferred_by_budget}, contexts deferred to the "
              f"next batch {cb.deferred_to_next_batch}", flush=True)
# static batches (the reference's rule: a batch runs until its longest row is done), same rows per batch, FIFO order
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 0
for b0 in range(0, N, B):
    chunk = reqs[b0:b0 + B]
    T0 = max(r[0].shape[0] for r in chunk)
    ids = torch.zeros(len(chunk), T0, 33, dtype=torch.long)
    mask = torch.zeros(len(chunk), T0, 33, dtype=chunk[0][1].dtype)
    for i, (ri, rm, _) in enumerate(chunk):
        ids[i, T0 - ri.shape[0]:], mask[i, T0 - ri.shape[0]:] = ri, rm
    n = max(r[2] for r in chunk)
    m.generate(ids.to(dev), mask.to(dev), max_new_frames=n, topk=1, stop_on_all_zeros=False)
    steps += n
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"static batches      : same utterances, {steps} frame-steps of {B} rows: {dt:.2f} s = {total / dt:.0f} useful frames/s", flush=True)
