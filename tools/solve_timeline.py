#!/usr/bin/env python3
"""Per-stream summary of the solve phase of the LAST bench step in a rocprofv3 --kernel-trace CSV:
    python3 tools/solve_timeline.py <run_kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

csv.field_size_limit(sys.maxsize)
rows = list(csv.DictReader(open(sys.argv[1])))


def nm(r):
    k = r["Kernel_Name"]
    i = k.find("gptq::")
    return k[i:i + 40].split("(")[0].split("<")[0] if i >= 0 else k.split("(")[0][-30:]


ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r), r["Stream_Id"], r["Queue_Id"]) for r in rows)
lasth = max(e[1] for e in ev if "hessian16" in e[2])
step = [e for e in ev if e[0] >= lasth]
t0 = step[0][0]
print(f"solve span {(max(e[1] for e in step) - t0) / 1e6:.3f} ms, {len(step)} kernels")
per = defaultdict(list)
for e in step:
    per[e[3]].append(e)
for st, l in sorted(per.items()):
    busy = sum(e[1] - e[0] for e in l)
    gaps = [l[i + 1][0] - l[i][1] for i in range(len(l) - 1)]
    print(f"stream {st} queue {l[0][4]}: {len(l)} kernels, busy {busy / 1e6:.2f} ms, span {(l[0][0] - t0) / 1e6:.2f}..{(l[-1][1] - t0) / 1e6:.2f} ms, "
          f"gaps {sum(g for g in gaps if g > 0) / 1e6:.2f} ms")
    by = defaultdict(lambda: [0, 0])
    for e in l:
        by[e[2]][0] += e[1] - e[0]
        by[e[2]][1] += 1
    for n, (t, c) in sorted(by.items(), key=lambda x: -x[1][0])[:8]:
        print(f"     {n:34s} {c:4d} x {t / c / 1e3:7.1f} us = {t / 1e6:6.2f} ms")
