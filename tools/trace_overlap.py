"""Summarise a rocprofv3 --kernel-trace CSV: per-stream busy time, wall span, and the concurrency profile.
Usage: python tools/trace_overlap.py <kernel_trace.csv> [name-substring-to-window-on]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else None
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:],
               r.get("Stream_Id", r.get("Queue_Id", "?")), r.get("Queue_Id", "?")))
ev.sort()
if key:
    sel = [e for e in ev if key in e[2]]
    lo, hi = sel[0][0], sel[-1][1]
    ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
t0, t1 = ev[0][0], max(e[1] for e in ev)
print(f"kernels {len(ev)}  span {(t1 - t0) / 1e6:.3f} ms")
per = defaultdict(lambda: [0, 0, None, None])
for s, e, n, st, q in ev:
    p = per[(st, q)]
    p[0] += e - s
    p[1] += 1
    p[2] = s if p[2] is None else min(p[2], s)
    p[3] = e if p[3] is None else max(p[3], e)
for k, p in sorted(per.items()):
    print(f"stream {k[0]:>4} queue {k[1]:>4}: {p[1]:5d} kernels busy {p[0] / 1e6:8.3f} ms  span {(p[2] - t0) / 1e6:8.3f}..{(p[3] - t0) / 1e6:8.3f} ms")
# concurrency histogram
pts = []
for s, e, *_ in ev:
    pts.append((s, 1))
    pts.append((e, -1))
pts.sort()
hist = defaultdict(int)
cur, last = 0, pts[0][0]
for t, d in pts:
    hist[cur] += t - last
    last = t
    cur += d
for c in sorted(hist):
    print(f"  {c} kernels in flight: {hist[c] / 1e6:8.3f} ms")
byname = defaultdict(lambda: [0, 0])
for s, e, n, *_ in ev:
    byname[n][0] += e - s
    byname[n][1] += 1
for n, (t, c) in sorted(byname.items(), key=lambda x: -x[1][0])[:12]:
    print(f"  {n:42s} {c:5d} x {t / c / 1e3:8.1f} us = {t / 1e6:8.3f} ms")
