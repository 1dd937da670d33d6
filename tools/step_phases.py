#!/usr/bin/env python3
"""Where one bench step goes, from a rocprofv3 --kernel-trace rocpd database: the Hessian and solve windows of the LAST
step's groups (a Hessian window = a run of hessian16 kernels; the solve window = what follows until the next group's
first Hessian kernel), the idle time inside each and the kernels that fill it.
    python3 tools/step_phases.py DIR/x_results.db [groups=4]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
ngroups = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = db.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: n.replace("void ", "").replace("gptq::", "").split("(")[0][:44]
is_h = lambda n: "hessian16" in n or "hessian_kernel" in n
# Hessian windows: maximal runs of Hessian kernels separated by less than 300 us of other kernels
wins, cur = [], None
for n, s, e in rows:
    if is_h(n):
        if cur and s - cur[1] < 1_500_000:
            cur[1] = max(cur[1], e)
        else:
            cur = [s, e]
            wins.append(cur)
wins = wins[-ngroups:]
t_end = rows[-1][2]
for gi, (hs, he) in enumerate(wins):
    nxt = wins[gi + 1][0] if gi + 1 < len(wins) else t_end
    seg = [(n, s, e) for n, s, e in rows if s >= he and s < nxt and not is_h(n)]
    busy_end, idle = he, 0
    for n, s, e in seg:
        if s > busy_end:
            idle += s - busy_end
        busy_end = max(busy_end, e)
    agg = defaultdict(lambda: [0, 0])
    for n, s, e in seg:
        agg[short(n)][0] += 1
        agg[short(n)][1] += e - s
    print(f"group {gi}: Hessian window {(he - hs) / 1e6:7.2f} ms, solve window {(nxt - he) / 1e6:7.2f} ms "
          f"(no kernel running {idle / 1e6:.2f} ms, {len(seg)} kernels)")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"      {t / 1e6:7.3f} ms {c:5d} x {k}")
