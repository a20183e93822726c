#!/usr/bin/env python
"""A/B helper: run bench.py once per engine-option set and print one compact line each.
usage: python tools/ab_bench.py [--steps N] [--batch B] [--extra "--topk 50"] "opt1=v,opt2=v" "opt=v" ...   ('' = defaults)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
steps, batch, extra = "100", "1", []
while args and args[0].startswith("--"):
    k = args.pop(0)
    v = args.pop(0)
    if k == "--steps":
        steps = v
    elif k == "--batch":
        batch = v
    elif k == "--extra":
        extra = v.split()
for spec in args or [""]:
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "10", "--lean", "--batch", batch] + extra
    for o in [x for x in spec.split(",") if x]:
        cmd += ["--opt", o]
    r = subprocess.run(cmd, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(f"{spec or 'default':40s} FAILED rc={r.returncode} {r.stderr[-300:]}", flush=True)
        continue
    d = json.loads(line[-1])
    ws = d.get("weight_streamer", {})
    par = d.get("parity", {})
    print(f"{spec or 'default':40s} {d['value']:9.2f} frames/s  {d['hip_event_ms_per_step']:.4f} ms/step  frac {d['roofline']['frac']:.4f}  "
          f"parity {par.get('equal_all')}  streamer gave_up={ws.get('gave_up')} fin={ws.get('finished')} late={ws.get('skipped_late_sample')} "
          f"segs={ws.get('segments')} sched={ws.get('scheduled_bytes', 0) / 1e6:.0f}MB of {ws.get('streamed_launch_bytes', 0) / 1e6:.0f}MB", flush=True)
