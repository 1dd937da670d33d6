#!/usr/bin/env python3
"""Kernel statistics of a rocprofv3 --kernel-trace run (rocpd sqlite database) as a small tracked CSV, in the layout of
rocprofv3's own kernel_stats.csv, plus -- with --dispatches SUBSTRING -- one line per dispatch of the kernels whose
name contains SUBSTRING (grid, start offset, duration), so that an average over ONE launch shape can be recomputed:
    python3 tools/kernel_stats_csv.py DIR/x_results.db profiles/rNN_bench_kernel_stats.csv [--dispatches hessian16_big16 OUT2.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                  "from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
with open(sys.argv[2], "w") as f:
    f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
    for n, c, t, a, mn, mx in rows:
        f.write(f'"{n}",{c},{t},{a:.1f},{100.0 * t / total:.2f},{mn},{mx}\n')
if "--dispatches" in sys.argv:
    i = sys.argv.index("--dispatches")
    sub, out = sys.argv[i + 1], sys.argv[i + 2]
    t0 = db.execute("select min(start) from kernels").fetchone()[0]
    with open(out, "w") as f:
        f.write('"Name","GridX","WorkgroupX","StartOffsetUs","DurationUs"\n')
        for n, gx, wx, s, e in db.execute("select name, grid_x, workgroup_x, start, end from kernels where name like ? order by start",
                                          (f"%{sub}%",)):
            f.write(f'"{n.split("(")[0]}",{gx},{wx},{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.2f}\n')
