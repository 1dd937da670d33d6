#!/usr/bin/env python3
"""Idle windows of the GPU (no kernel running on any stream) in a rocprofv3 rocpd database, for the last N ms:
    python3 tools/idle_gaps.py DIR/x_results.db [--last-ms 200] [--min-us 30]
Prints the total idle time and the largest gaps with the kernels before / after them."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
last_ms = float(sys.argv[sys.argv.index("--last-ms") + 1]) if "--last-ms" in sys.argv else 200.0
min_us = float(sys.argv[sys.argv.index("--min-us") + 1]) if "--min-us" in sys.argv else 30.0
rows = db.execute("select name, start, end from kernels order by start").fetchall()
t_end = max(r[2] for r in rows)
rows = [r for r in rows if r[1] >= t_end - last_ms * 1e6]
short = lambda n: n.split("(")[0].replace("void ", "").replace("gptq::", "")[:40]
cur_end, prev = rows[0][2], rows[0]
gaps, idle = [], 0.0
for r in rows[1:]:
    if r[1] > cur_end:
        g = (r[1] - cur_end) / 1e3
        idle += g
        if g >= min_us:
            gaps.append((g, short(prev[0]), short(r[0]), (cur_end - rows[0][1]) / 1e6))
    if r[2] > cur_end:
        cur_end, prev = r[2], r
print(f"window {(t_end - rows[0][1]) / 1e6:.1f} ms, idle {idle / 1e3:.2f} ms, gaps >= {min_us} us: {len(gaps)} totalling {sum(g[0] for g in gaps) / 1e3:.2f} ms")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"  {g[0]:8.1f} us at +{g[3]:7.2f} ms   {g[1]} -> {g[2]}")
