#!/usr/bin/env python3
"""Per-stream picture of the LAST bench step in a rocprofv3 --kernel-trace database (rocpd sqlite):
    python3 tools/step_streams.py RESULTS.db [steps_in_run]
prints, for every HIP stream, when it was busy inside the step and with which kernels, plus a coarse occupancy timeline
(one row per stream, one column per millisecond: share of that millisecond the stream had a kernel running)."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = db.execute("select start, end, name, stream_id from kernels order by start").fetchall()
t_first, t_last = ev[0][0], max(e[1] for e in ev)
# the run is warmup + steps equal steps: take the last 1/nsteps of the busy span
lo = t_last - (t_last - t_first) / nsteps
# snap to the first big Hessian launch after lo
for s, e, n, st in ev:
    if s >= lo and "hessian16_big16_kernel" in n:
        lo = s
        break
step = [(s, e, n.split("(")[0].replace("void ", "").replace("gptq::", "")[:44], st) for s, e, n, st in ev if s >= lo]
t0, t1 = step[0][0], max(e[1] for e in step)
print(f"last step: {(t1 - t0) / 1e6:.2f} ms, {len(step)} kernels")
per = defaultdict(list)
for e in step:
    per[e[3]].append(e)
nms = int((t1 - t0) / 1e6) + 1
for st, l in sorted(per.items(), key=lambda kv: kv[1][0][0]):
    busy = sum(e[1] - e[0] for e in l)
    print(f"stream {st}: {len(l)} kernels, busy {busy / 1e6:.2f} ms, {(l[0][0] - t0) / 1e6:.2f} .. {(max(e[1] for e in l) - t0) / 1e6:.2f} ms")
    by = defaultdict(lambda: [0, 0])
    for e in l:
        by[e[2]][0] += e[1] - e[0]
        by[e[2]][1] += 1
    for n, (t, c) in sorted(by.items(), key=lambda x: -x[1][0])[:6]:
        print(f"     {n:44s} {c:4d} x {t / c / 1e3:8.1f} us = {t / 1e6:6.2f} ms")
    occ = [0.0] * nms
    for s, e, _, _ in l:
        a, b = (s - t0) / 1e6, (e - t0) / 1e6
        for m in range(int(a), min(int(b) + 1, nms)):
            occ[m] += max(0.0, min(b, m + 1) - max(a, m))
    print("     " + "".join(" .:-=+*#%@"[min(9, int(o * 10))] for o in occ))
