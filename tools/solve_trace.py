#!/usr/bin/env python3
"""Timeline of ONE solve from a rocprofv3 kernel trace (rocpd sqlite database, `rocprofv3 --kernel-trace -d DIR`):
    python3 tools/solve_trace.py DIR/*_results.db [--solve N] [--list]
Picks the N-th last gptq_fasterquant call (from its dead_fix_kernel to its sum_kernel), and prints per kernel name:
launches, busy time, and -- for the stream the solve was enqueued on -- the idle time between kernels, i.e. where the
critical path is not a kernel."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, stream_id, start, end, grid_x / workgroup_x, grid_y from kernels order by start"))
short = lambda n: n.split("(")[0].replace("void ", "").replace("gptq::", "")[:44]
starts = [i for i, r in enumerate(rows) if "dead_fix_kernel" in r[0]]
ends = [i for i, r in enumerate(rows) if "sum_kernel" in r[0]]
which = int(sys.argv[sys.argv.index("--solve") + 1]) if "--solve" in sys.argv else 1
i0 = starts[-which]
i1 = min(e for e in ends if e > i0)
sel = rows[i0:i1 + 1]
t0, t1 = sel[0][2], sel[-1][3]
main = sel[0][1]
print(f"solve window {(t1 - t0) / 1e6:.3f} ms, {len(sel)} kernels, main stream {main}")
agg = collections.OrderedDict()
for n, st, a, b, gx, gy in sel:
    k = (short(n), "main" if st == main else "side")
    d = agg.setdefault(k, [0, 0.0])
    d[0] += 1
    d[1] += (b - a) / 1e3
for (n, st), (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:44s} {st:4s} n={cnt:5d} busy {us / 1e3:8.3f} ms  avg {us / cnt:8.1f} us")
mains = [r for r in sel if r[1] == main]
busy = sum(b - a for _, _, a, b, _, _ in mains) / 1e6
gaps = collections.defaultdict(lambda: [0, 0.0])
prev = None
for r in mains:
    if prev is not None:
        g = (r[2] - prev[3]) / 1e3
        k = f"{short(prev[0])} -> {short(r[0])}"
        gaps[k][0] += 1
        gaps[k][1] += max(g, 0.0)
    prev = r
print(f"main stream: busy {busy:.3f} ms, idle {(t1 - t0) / 1e6 - busy:.3f} ms")
for k, (cnt, us) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  gap {k:80s} n={cnt:4d} total {us / 1e3:7.3f} ms avg {us / cnt:6.1f} us")
if "--list" in sys.argv:
    for n, st, a, b, gx, gy in sel[:400]:
        print(f"{(a - t0) / 1e3:10.1f} us +{(b - a) / 1e3:8.1f}  {'M' if st == main else 's'} {short(n)} grid {gx}x{gy}")
