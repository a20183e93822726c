#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel and per-(kernel,shape) stats.
usage: python tools/rocprof_summary.py results.db [frames]  -> markdown on stdout"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
frames = float(sys.argv[2]) if len(sys.argv) > 2 else None
cur = db.cursor()
rows = list(cur.execute("select name, count(*), sum(end-start)/1000.0, avg(end-start)/1000.0, min(end-start)/1000.0,"
                        " max(end-start)/1000.0 from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot/1000:.2f} ms over {sum(r[1] for r in rows)} dispatches\n")
print("| calls | total ms | avg us | min us | max us | % | kernel |")
print("|---|---|---|---|---|---|---|")
for r in rows[:30]:
    print(f"| {r[1]} | {r[2]/1000:.2f} | {r[3]:.2f} | {r[4]:.2f} | {r[5]:.2f} | {100*r[2]/tot:.1f} | `{r[0][:100]}` |")
print("\nper (kernel, grid, LDS) for the engine's kernels:\n")
print("| kernel | grid (threads) | LDS B | VGPR | calls | calls/frame | avg us | min us |")
print("|---|---|---|---|---|---|---|---|")
q = ("select name, grid_x, lds_size, vgpr_count, count(*), avg(end-start)/1000.0, min(end-start)/1000.0 from kernels "
     "where name like '%gemv_kernel%' or name like '%attn%' or name like 'sample%' or name like '%embed%' "
     "or name like '%gemm_%' or name like '%rmsnorm%' or name like '%rope%' group by name, grid_x, lds_size order by name, grid_x")
for r in cur.execute(q):
    cpf = f"{r[4]/frames:.1f}" if frames else ""
    print(f"| `{r[0][:70]}` | {r[1]} | {r[2]} | {r[3]} | {r[4]} | {cpf} | {r[5]:.2f} | {r[6]:.2f} |")

# ---- frame-step consistency: kernels between two consecutive decode embed_sum launches ----------------------
ks = list(cur.execute("select start, end, name, grid_x from kernels order by start"))
idx = [i for i, k in enumerate(ks) if "embed_sum" in k[2] and k[3] <= 1024]
if len(idx) >= 3:
    spans, sums, counts = [], [], []
    for a_, b_ in zip(idx[-4:-1], idx[-3:]):
        fr = ks[a_:b_]
        spans.append((fr[-1][1] - fr[0][0]) / 1e6)
        sums.append(sum(k[1] - k[0] for k in fr) / 1e6)
        counts.append(len(fr))
    print(f"\nframe-step consistency (last {len(spans)} steps, UNDER THE PROFILER): {counts[0]} launches per step, "
          f"sum of kernel durations {sum(sums)/len(sums):.3f} ms, first-start-to-last-end span {sum(spans)/len(spans):.3f} ms "
          f"(bench.py's un-profiled HIP-event time per step is reported in profiles/r01_bench.json)")
