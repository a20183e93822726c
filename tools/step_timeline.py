#!/usr/bin/env python
"""Launch-by-launch anatomy of the LAST full frame-step in a rocprofv3 kernel trace (rocpd sqlite): for every distinct
(kernel, grid, workgroup) of the step: launches, mean duration (start -> end), mean gap to the NEXT launch's start
(end -> next start: the dependent-launch boundary) and mean start-to-start cost, i.e. what the launch costs in the chain.
usage: python tools/step_timeline.py results.db [--dump N]  -> markdown on stdout"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
gy = "grid_y" if "grid_y" in cols else "0"
wg = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else "0")
ks = list(cur.execute(f"select start, end, name, grid_x, {gy}, {wg} from kernels order by start"))
# the decode step's embedding launch = the smallest embed_sum grid of the trace (prefill launches cover whole contexts)
es = [k[3] * max(1, k[4]) for k in ks if "embed_sum" in k[2]]
idx = [i for i, k in enumerate(ks) if "embed_sum" in k[2] and k[3] * max(1, k[4]) == min(es)]
if len(idx) < 3:
    sys.exit("no frame-steps found")
a, b = idx[-3], idx[-2]
step = ks[a:b + 1]
agg = collections.OrderedDict()
for i in range(len(step) - 1):
    s, e, name, gx, gyv, wgx = step[i]
    key = (name.split("(")[0][:70], gx, gyv, wgx)
    d = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
    d[0] += 1
    d[1] += (e - s) / 1e3
    d[2] += (step[i + 1][0] - e) / 1e3
    d[3] += (step[i + 1][0] - s) / 1e3
span = (step[-1][0] - step[0][0]) / 1e6
print(f"one frame-step: {len(step) - 1} launches, first start -> next step's first start {span:.3f} ms (under the profiler)\n")
print("| kernel | grid x,y (threads) | wg | launches | dur us | gap us | start-to-start us | ms/step |")
print("|---|---|---|---|---|---|---|---|")
for (name, gx, gyv, wgx), d in sorted(agg.items(), key=lambda kv: -kv[1][3]):
    n = d[0]
    print(f"| `{name}` | {gx},{gyv} | {wgx} | {n} | {d[1]/n:.2f} | {d[2]/n:.2f} | {d[3]/n:.2f} | {d[3]/1e3:.3f} |")
if "--dump" in sys.argv:
    n = int(sys.argv[sys.argv.index("--dump") + 1])
    print("\nfirst launches of the step (us from step start: start, dur, name, grid):\n")
    t0 = step[0][0]
    for s, e, name, gx, gyv, wgx in step[:n]:
        print(f"{(s - t0)/1e3:9.2f} {(e - s)/1e3:7.2f}  {name.split('(')[0][:60]}  {gx},{gyv}/{wgx}")
if "--gaps" in sys.argv:
    # every boundary of the LAST FOUR steps whose end -> next-start gap exceeds T us: position in the step, the two kernels
    thr = float(sys.argv[sys.argv.index("--gaps") + 1])
    print(f"\nboundaries with a gap above {thr} us (step, index in step, gap us, kernel that ended -> kernel that started):\n")
    for si in range(max(0, len(idx) - 5), len(idx) - 1):
        st = ks[idx[si]:idx[si + 1] + 1]
        for i in range(len(st) - 1):
            g = (st[i + 1][0] - st[i][1]) / 1e3
            if g > thr:
                print(f"step {si:3d} #{i:4d} gap {g:8.2f}  {st[i][2].split('(')[0][:58]} {st[i][3]}/{st[i][5]} -> "
                      f"{st[i + 1][2].split('(')[0][:58]} {st[i + 1][3]}/{st[i + 1][5]}")
