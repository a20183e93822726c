#!/usr/bin/env python
"""Weight-streamer probe: for every streamed launch of the B=1 frame-step, (a) does workgroup b run on XCD (b + rot) % 8
as the streamer assumes, (b) how many shader clocks pass from kernel entry until the weights have been consumed, with the
streamer off and on (an L2 hit returns in a few hundred clocks, an HBM miss in ~1-2 thousand).
usage: python tools/streamer_probe.py [engine opt=value ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the probe is compiled in only in the -DCSM_PROBE variant of the library (even a disabled probe costs 6 % per frame)
PROBE_LIB = os.environ.get("CSM_PROBE_LIB") or os.path.join(ROOT, "csm-hf_amd", "libcsm_hip_probe.so")
from csm_hf_amd.build import build_library, LIB, _sources_mtime  # noqa: E402
if not os.environ.get("CSM_PROBE_LIB") and (not os.path.exists(PROBE_LIB) or os.path.getmtime(PROBE_LIB) < _sources_mtime()):
    build_library(force=True, defines=("CSM_PROBE",), out=PROBE_LIB)   # stale or missing: rebuild (hipcc, ~2 min)
os.environ["CSM_HIP_LIB"] = PROBE_LIB
from csm_hf_amd import CSMConfig, CSMModel  # noqa: E402
from csm_hf_amd.synth import synth_state_dict, synth_context  # noqa: E402

dev = torch.device("cuda:0")
cfg = CSMConfig()
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev, bf16_representable=True)
m = CSMModel(cfg)
m.load_state_dict(sd)
del sd
ids, mask = synth_context(cfg, 1, 128, 384, seed=2)
eng = m._ensure_engine(1, 512 + 64, 64, 512)
TOPK = 1
for o in sys.argv[1:]:
    k, v = o.split("=")
    if k == "topk":          # sampled path (top-k, T = 0.9) instead of greedy
        TOPK = int(v)
        continue
    eng.set_option(k, int(v))
NL = 800
res = {}
for mode in (0, 1):
    eng.set_option("weight_prefetch", mode)
    buf = torch.zeros(NL, 2048, 2, dtype=torch.int32, device=dev)
    eng.set_debug_buffer(buf, NL)
    eng.reset()
    eng.set_kv_start([0])
    eng.prefill(ids, mask, want_outputs=False)
    s = eng.sampling(temperature=1.0 if TOPK == 1 else 0.9, topk=TOPK)
    eng.generate(s, 8, True)
    eng.sync()
    res[mode] = buf.cpu().clone()
    st = eng.prefetch_stats()
geoms = eng.last_geoms()
rot = st["xcd_rotation"]
print("streamer stats:", st)
bad = 0
rows = {}
for i, (N, K, grid, tpb, kind) in enumerate(geoms):
    if i >= NL:
        break
    g = min(grid, 2048)
    x = res[1][i, :g, 0] & 15
    exp = (torch.arange(g) + rot) % 8
    mism = int((x != exp).sum())
    bad += mism
    key = (N, K, grid, kind)
    t0 = res[0][i, :g, 1].float()
    t1 = res[1][i, :g, 1].float()
    x0 = (res[0][i, :g, 0] >> 8).float()
    x1 = (res[1][i, :g, 0] >> 8).float()
    rows.setdefault(key, []).append((float(t0.mean()), float(t0.max()), float(t1.mean()), float(t1.max()), mism, float(x0.mean()), float(x1.mean())))
print(f"launches {len(geoms)}, workgroups not on XCD (b + {rot}) % 8: {bad}")
print("N, K, grid, kind : launches | clocks until weights consumed, mean / max over workgroups: streamer off -> on")
for key, v in rows.items():
    n = len(v)
    a = [sum(x[j] for x in v) / n for j in range(4)]
    xa = [sum(x[j] for x in v) / n for j in (5, 6)]
    print(f"{key}: {n:4d} | off mean {a[0]:7.0f} max {a[1]:7.0f} (activations in at {xa[0]:6.0f}) -> on mean {a[2]:7.0f} max {a[3]:7.0f} (activations in at {xa[1]:6.0f})   mismatched wgs {sum(x[4] for x in v)}")

# per-launch sequence of the first two decoder passes + octiles (by workgroup index) of two steady-state launches
print("\nfirst 48 streamed launches: (N, K) mean clocks off -> on")
for i in range(min(48, len(geoms))):
    N, K, grid, tpb, kind = geoms[i]
    g = min(grid, 2048)
    print(f"  {i:3d} ({N:5d},{K:5d}) grid {grid:4d}: {float(res[0][i, :g, 1].float().mean()):7.0f} -> {float(res[1][i, :g, 1].float().mean()):7.0f}"
          f"   activations in at {float((res[0][i, :g, 0] >> 8).float().mean()):6.0f} -> {float((res[1][i, :g, 0] >> 8).float().mean()):6.0f}"
          f"   min/max over workgroups (on) {int(res[1][i, :g, 1].min())}/{int(res[1][i, :g, 1].max())}")
for shape in ((16384, 1024), (1024, 8192), (2048, 8192), (1536, 1024)):
    idx = [i for i, gm in enumerate(geoms) if (gm[0], gm[1]) == shape and i < NL]
    if len(idx) < 12:
        idx = idx[-1:] if idx else []
    else:
        idx = idx[40:42]
    for i in idx:
        g = min(geoms[i][2], 2048)
        for mode in (0, 1):
            t = res[mode][i, :g, 1].float()
            oct_ = [float(c.mean()) for c in t.chunk(8)]
            print(f"  launch {i} {shape} streamer {'on ' if mode else 'off'}: octiles of workgroup index " + " ".join(f"{o:6.0f}" for o in oct_))
