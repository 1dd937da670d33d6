#!/usr/bin/env python3
"""Per-kernel launch counts and average / min durations from a rocprofv3 rocpd database:
    python3 tools/kernel_times.py DIR/*_results.db [substring ...]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
subs = sys.argv[2:]
rows = db.execute("select name, count(*), avg(end - start) / 1e3, min(end - start) / 1e3, sum(end - start) / 1e6 from kernels "
                  "group by name order by 5 desc").fetchall()
print("calls  avg_us  min_us  total_ms  name")
for n, c, a, m, t in rows:
    if subs and not any(s in n for s in subs):
        continue
    print(f"{c:5d} {a:8.2f} {m:8.2f} {t:9.3f}  {n[:100]}")
