#!/usr/bin/env python
"""In-step timeline of one frame-step WITH the weight streamer running (rocprofv3 serialises / perturbs the replay; this does
not): the -DCSM_TIMELINE build of the library records, per workgroup of every decode-path launch, the 100 MHz constant clock at
kernel entry and after the workgroup's last store.  Per launch: first / last workgroup start, first / last workgroup end; from
these the gap in front of the launch (previous launch's last end -> this launch's first start: the boundary), the dispatch ramp
(first -> last start) and the body (first start -> last end).  Printed per launch kind (kernel family, grid) in step order, with
the closed-form budget: sum(gap + body) against the HIP-event step time of the product build.

usage: python tools/b1_timeline.py [--batch B] [--md out.md] [engine opt=value ...]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--ctx", type=int, default=512)
ap.add_argument("--frames", type=int, default=12)
ap.add_argument("--topk", type=int, default=1)
ap.add_argument("--md", default=None)
ap.add_argument("--json", default=None, help="write the streamer-on per-launch-kind table as JSON (bench.py attaches profiles/launch_kinds_b<B>.json to its record)")
ap.add_argument("opts", nargs="*")
a = ap.parse_args()

TL_LIB = os.environ.get("CSM_TL_LIB") or os.path.join(ROOT, "csm-hf_amd", "libcsm_hip_timeline.so")
from csm_hf_amd.build import build_library, _sources_mtime  # noqa: E402
if not os.environ.get("CSM_TL_LIB") and (not os.path.exists(TL_LIB) or os.path.getmtime(TL_LIB) < _sources_mtime()):
    build_library(force=True, defines=("CSM_TIMELINE",), out=TL_LIB)
os.environ["CSM_HIP_LIB"] = TL_LIB
from csm_hf_amd import CSMConfig, CSMModel  # noqa: E402
from csm_hf_amd.synth import synth_state_dict, synth_context  # noqa: E402

KINDS = {2: "attn_oproj", 3: "attn_decode", 4: "attn_combine", 5: "sample", 6: "embed_sum", 7: "attn_oproj_gqa", 8: "attn_decode_gqa"}
PRO = {0: "plain", 1: "norm", 3: "toknorm", 2: "combine", 4: "sample"}
EPI = {0: "store", 1: "resid", 2: "swiglu", 3: "qkv", 4: "argmax"}


def kind_name(k):
    if k in KINDS:
        return KINDS[k]
    fam = "gemv1" if 0x10 <= k < 0x40 else "gemv_lds" if 0x40 <= k < 0x70 else "gemm16" if 0x70 <= k < 0xa0 else f"kind{k}"
    v = k - (0x10 if k < 0x40 else 0x40 if k < 0x70 else 0x70)
    return f"{fam}<{PRO.get(v >> 3, v >> 3)},{EPI.get(v & 7, v & 7)}>"


dev = torch.device("cuda:0")
cfg = CSMConfig()
sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev, bf16_representable=True)
m = CSMModel(cfg)
m.load_state_dict(sd)
del sd
B = a.batch
ids, mask = synth_context(cfg, B, a.ctx // 4, a.ctx - a.ctx // 4, seed=2)
eng = m._ensure_engine(B, a.ctx + 64, 64, B * a.ctx)
for o in a.opts:
    k, v = o.split("=")
    eng.set_option(k, int(v))
NL = 900
out_lines = []


def emit(s=""):
    print(s)
    out_lines.append(s)


for streamer in (1, 0):
    eng.set_option("weight_prefetch", streamer)
    buf = torch.zeros(NL, 2048, 2, dtype=torch.int32, device=dev)
    eng.set_debug_buffer(buf, NL)
    eng.reset()
    eng.set_kv_start([0] * B)
    eng.prefill(ids, mask, want_outputs=False)
    s = eng.sampling(temperature=1.0, topk=a.topk, seed=3)
    eng.generate(s, a.frames, True)
    eng.sync()
    ms = eng.last_generate_ms() / a.frames
    raw = buf.cpu().numpy().astype("int64") & 0xffffffff
    rows = []
    for i in range(NL):
        tag = int(raw[i, 2047, 0])
        if tag == 0:
            break
        kind, grid = tag & 0xff, tag >> 8
        g = min(grid, 2047)
        st, en = raw[i, :g, 0], raw[i, :g, 1]
        rows.append((kind, grid, int(st.min()), int(st.max()), int(en.min()), int(en.max())))
    n = len(rows)
    if a.md and streamer == 1:   # per-workgroup durations of the first launch of every (kind, grid): where inside a launch the time goes
        import numpy as np
        seen = set()
        with open(a.md.replace(".md", "_per_wg.txt"), "w") as f:
            for i, (kind, grid, s0, s1, e0, e1) in enumerate(rows[40:], 40):
                if (kind, grid) in seen:
                    continue
                seen.add((kind, grid))
                g = min(grid, 2047)
                st, en = raw[i, :g, 0] - s0, raw[i, :g, 1] - s0
                d = (en - st) / 100.0
                q = lambda x, p: float(np.percentile(x, p))
                f.write(f"{kind_name(kind)} wgs {grid}: start offset us p0/p50/p100 {q(st,0)/100:.2f}/{q(st,50)/100:.2f}/{q(st,100)/100:.2f}  "
                        f"per-wg duration us p0/p10/p50/p90/p100 {q(d,0):.2f}/{q(d,10):.2f}/{q(d,50):.2f}/{q(d,90):.2f}/{q(d,100):.2f}  end p50/p100 {q(en,50)/100:.2f}/{q(en,100)/100:.2f}\n")
    if a.md:   # raw per-launch rows for offline analysis: kind, workgroups, first start, last start, first end, last end (10 ns ticks from the step's first start)
        with open(a.md.replace(".md", f"_raw_streamer{streamer}.csv"), "w") as f:
            for (kind, grid, s0, s1, e0, e1) in rows:
                f.write(f"{kind_name(kind)},{grid},{s0 - rows[0][2]},{s1 - rows[0][2]},{e0 - rows[0][2]},{e1 - rows[0][2]}\n")
    emit(f"\n## streamer {'on' if streamer else 'off'}: {n} launches in the last replayed frame-step; step time of this (probe) build {ms * 1e3:.1f} us by HIP events "
         f"(product build: see bench); clock = s_memrealtime, 10 ns ticks")
    t_first, t_last = rows[0][2], rows[-1][5]
    emit(f"first workgroup start -> last workgroup end of the step: {(t_last - t_first) / 100:.1f} us")
    agg = {}
    order = []
    prev_end = None
    tot_gap = tot_body = 0.0
    for (kind, grid, s0, s1, e0, e1) in rows:
        gap = (s0 - prev_end) / 100 if prev_end is not None else 0.0
        prev_end = e1
        key = (kind, grid)
        if key not in agg:
            agg[key] = [0, 0.0, 0.0, 0.0, 0.0]
            order.append(key)
        r = agg[key]
        r[0] += 1
        r[1] += gap
        r[2] += (s1 - s0) / 100
        r[3] += (e1 - s0) / 100
        r[4] += (e1 - e0) / 100
        tot_gap += gap
        tot_body += (e1 - s0) / 100
    emit("| kernel | workgroups | launches | gap in front us | start ramp us | body us (first start -> last end) | end spread us | gap + body us | per step us |")
    emit("|---|---|---|---|---|---|---|---|---|")
    for key in order:
        c, g, rmp, body, spr = agg[key]
        emit(f"| `{kind_name(key[0])}` | {key[1]} | {c} | {g / c:.2f} | {rmp / c:.2f} | {body / c:.2f} | {spr / c:.2f} | {(g + body) / c:.2f} | {g + body:.1f} |")
    emit(f"| total | | {n} | {tot_gap / max(n - 1, 1):.2f} | | {tot_body / n:.2f} | | | {tot_gap + tot_body:.1f} |")
    if a.json and streamer == 1:
        import hashlib
        import json
        import subprocess
        from csm_hf_amd.build import LIB
        kinds = [{"kernel": kind_name(k[0]), "workgroups": k[1], "launches": agg[k][0], "gap_us": round(agg[k][1] / agg[k][0], 2),
                  "body_us": round(agg[k][3] / agg[k][0], 2), "us_per_step": round(agg[k][1] + agg[k][3], 1)} for k in order]
        rec = {"batch": B, "ctx": a.ctx, "topk": a.topk, "launches_per_step": n, "step_us_probe_build": round(tot_gap + tot_body, 1),
               "how": "tools/b1_timeline.py: per-workgroup s_memrealtime at entry / after the last store in the -DCSM_TIMELINE build of the SAME sources "
                      "(round 6: 15-25 % slower than the product build -- the probe's s_memrealtime at kernel entry is a scalar-memory wait in front of the first load, exactly what the kernel-argument preload removed from the product; proportions between launch kinds hold, absolute bodies are ~0.5 us long), streamer on; gap = previous launch's last end -> first start, body = first start -> last end",
               "src_sha256": __import__("csm_hf_amd.build", fromlist=["sources_sha256"]).sources_sha256(),
               "commit": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip(),
               "kinds": kinds}
        json.dump(rec, open(a.json, "w"), indent=1)
    eng.set_debug_buffer(None, 0)

if a.md:
    with open(a.md, "w") as f:
        f.write("# `python tools/b1_timeline.py " + " ".join(sys.argv[1:]) + "`\n")
        f.write("\n".join(out_lines) + "\n")
