#!/usr/bin/env python
"""Per-kernel sums of the counters in the rocprofv3 result databases under a directory: python tools/pmc_kernels.py <dir>
(one sub-directory per --pmc pass, as tools/pmc_prefill.sh writes them).  Prints one table per pass."""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
for db_path in sorted(glob.glob(os.path.join(root, "*", "*_results.db"))):
    db = sqlite3.connect(db_path)
    try:
        rows = db.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events "
                          "where name not like '%at::native%' and name not like '%rocclr%' group by name, counter_name").fetchall()
    except sqlite3.Error as ex:
        print(db_path, "unreadable:", ex)
        continue
    print("##", os.path.basename(os.path.dirname(db_path)))
    per = {}
    for name, cn, n, v in rows:
        per.setdefault(name, {})[cn] = (n, v)
    for name, d in sorted(per.items(), key=lambda kv: -max(x[1] for x in kv[1].values())):
        n = max(x[0] for x in d.values())
        print(f"{name[:78]:78s} n={n:5d} " + "  ".join(f"{c}={v / n:.4g}/launch" for c, (n_, v) in sorted(d.items())))
