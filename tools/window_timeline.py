"""Kernels of the LAST step of a rocprofv3 kernel trace (rocpd sqlite) between the last launch of kernel-prefix A and the first launch of kernel-prefix B after it, with queue
(= HIP stream), start offset and duration; plus how much of the window has NO kernel running and how much only small ones (dev tooling).
   window_timeline.py results.db A B"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
a, b = sys.argv[2], sys.argv[3]
rows = con.execute("select name, start, end, queue_id from kernels order by start").fetchall()
rows = [(n.replace("(anonymous namespace)::", "").replace("void ", ""), s, e, q) for n, s, e, q in rows]
sq = [i for i, r in enumerate(rows) if r[0].startswith("sqnorm_kernel")]
lo, hi = sq[-2], sq[-1]
step = rows[lo:hi]
ia = max(i for i, r in enumerate(step) if r[0].startswith(a) and any(x[0].startswith(b) for x in step[i:]))
ib = next(i for i in range(ia, len(step)) if step[i][0].startswith(b))
t0, t1 = step[ia][2], step[ib][1]
sel = [r for r in step if r[2] > t0 and r[1] < t1]
qs = sorted({r[3] for r in sel})
print(f"window {a} -> {b}: {(t1 - t0) / 1e3:.1f} us, {len(sel)} kernels on queues {qs}")
for n, s, e, q in sel:
    print(f"  q{qs.index(q)}  {(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {n[:100]}")
ev = sorted([(max(s, t0), 1) for _, s, e, _ in sel] + [(min(e, t1), -1) for _, s, e, _ in sel])
busy, depth, last = 0, 0, t0
for t, d in ev:
    if depth > 0:
        busy += t - last
    depth += d
    last = t
print(f"some kernel running: {busy / 1e3:.1f} us of {(t1 - t0) / 1e3:.1f}")
