#!/usr/bin/env python
"""Per-launch time of the decode GEMV kernel for the csm-1b shapes, measured as a dependent chain in a
hipGraph (tools: csm_bench_gemv).  Prints a markdown table; compare with tools/ubench/chain (raw floor)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from csm_hf_amd import CSMConfig
from csm_hf_amd.engine import Engine
from csm_hf_amd.synth import synth_state_dict

cfg = CSMConfig.tiny()
eng = Engine(cfg, synth_state_dict(cfg), "cuda:0", torch.float32, max_batch=4, max_len=64, max_frames=4, max_prefill_rows=128)
shapes = [("dec qkv", 1536, 1024, True, 0), ("dec o", 1024, 1024, False, 1), ("dec gate/up", 16384, 1024, True, 2),
          ("dec down", 1024, 8192, False, 1), ("bb qkv", 3072, 2048, True, 0), ("bb o", 2048, 2048, False, 1),
          ("bb gate/up", 16384, 2048, True, 2), ("bb down", 2048, 8192, False, 1), ("audio head", 2051, 1024, True, 0),
          ("c0 head+proj", 3075, 2048, True, 0)]
Ms = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1]
kw = {k: int(v) for k, v in (a.split("=") for a in sys.argv[1:] if "=" in a and not a.startswith("opt:"))}
# MFMA overrides: g16=NW,KB,PT  (packed into grid_cap's upper bits)
if "g16" in "".join(sys.argv):
    nw, kb, pt = [int(v) for v in [a for a in sys.argv if a.startswith("g16:")][0][4:].split(",")]
    kw["grid_cap"] = (nw << 16) | (kb << 24) | ((1 << 30) if pt == 4 else 0)
only = [a[5:] for a in sys.argv if a.startswith("only:")]
for a in sys.argv:   # engine options, e.g. opt:tile_weights=0
    if a.startswith("opt:"):
        k_, v_ = a[4:].split("=")
        eng.set_option(k_, int(v_))
print("| shape | N | K | MB | " + " | ".join(f"M={m} us (TB/s)" for m in Ms) + " |")
print("|---|---|---|---|" + "---|" * len(Ms))
for name, N, K, norm, epi in shapes:
    if only and not any(o in name for o in only):
        continue
    cells = []
    for M in Ms:
        us, wb = eng.bench_gemv(N, K, M=M, norm=norm, epi=epi, **kw)
        cells.append(f"{us:.2f} ({wb / us / 1e6:.2f})")
    print(f"| {name} | {N} | {K} | {N * K * 2 / 1e6:.1f} | " + " | ".join(cells) + " |")
