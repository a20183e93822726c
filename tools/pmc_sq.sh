#!/bin/bash
# SQ wave-state counters per kernel for a short bench run (one --pmc pass, kernel-trace only).
# usage: bash tools/pmc_sq.sh "<bench.py args>" outdir
A=${1:---batch 16}; O=${2:-gpurun_out/sq}; R=$PWD; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES \
  -d $R/$O/db -o r -- python $R/bench.py --steps 3 --warmup 1 --lean $A > $R/$O/log.txt 2>&1
echo "rc=$?"
cd $R
python - <<'PY' $O
import sqlite3, sys, collections
o = sys.argv[1]
db = sqlite3.connect(f"{o}/db/r_results.db")
rows = db.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events where name like '%gemm16%' or name like '%gemv%' or name like '%attn_%' or name like 'sample%' group by name, counter_name").fetchall()
t = collections.defaultdict(dict)
for n, c, k, v in rows:
    t[n][c] = (k, v)
print("| kernel | launches | waves | wave-cycles/wave (quad) | parked % | issue-stall % | active % | VALU-active % | VALU insts/wave |")
print("|---|---|---|---|---|---|---|---|---|")
for n, d in sorted(t.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", (0, 0))[1]):
    if "SQ_WAVES" not in d: continue
    k = d["SQ_WAVES"][0]; w = d["SQ_WAVES"][1] or 1; wc = d["SQ_WAVE_CYCLES"][1] or 1
    print(f"| `{n[:70]}` | {k} | {w/k:.0f} | {wc/w:.0f} | {100*d['SQ_WAIT_ANY'][1]/wc:.0f} | {100*d['SQ_WAIT_INST_ANY'][1]/wc:.0f} | "
          f"{100*d['SQ_ACTIVE_INST_ANY'][1]/wc:.0f} | {100*d['SQ_ACTIVE_INST_VALU'][1]/wc:.0f} | {d['SQ_INSTS_VALU'][1]/w:.0f} |")
PY
rm -rf $O/db
