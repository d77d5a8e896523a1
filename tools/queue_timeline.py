"""Per-queue view of the last step in a rocprofv3 kernel trace (rocpd sqlite), for any task (dev tooling): wall time of the step, busy time and kernel count of every
hardware queue (= HIP stream), idle gaps of the busiest queue by (kernel before -> kernel after), and the per-kernel totals of each queue.
   queue_timeline.py results.db [marker-kernel-prefix]      the step boundary is the last launch of the marker kernel (default adamw_kernel) in a run of them"""
import sqlite3
import sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "adamw_kernel"
rows = con.execute("select name, start, end, queue_id from kernels order by start").fetchall()
rows = [(n.replace("(anonymous namespace)::", "").replace("void ", ""), s, e, q) for n, s, e, q in rows]
mk = [i for i, r in enumerate(rows) if r[0].startswith(marker)]
ends = [i for j, i in enumerate(mk) if j + 1 == len(mk) or mk[j + 1] - i > 8]
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]
t0, t1 = step[0][1], max(r[2] for r in step)
print(f"step: {len(step)} kernels, wall {(t1 - t0) / 1e6:.2f} ms, kernel-time sum {sum(r[2] - r[1] for r in step) / 1e6:.2f} ms")
byq = defaultdict(list)
for r in step:
    byq[r[3]].append(r)
for q, rs in sorted(byq.items()):
    print(f"  queue {q}: {len(rs):4d} kernels, busy {sum(e - s for _, s, e, _ in rs) / 1e6:6.2f} ms, active {(rs[0][1] - t0) / 1e6:.2f} .. {(max(r[2] for r in rs) - t0) / 1e6:.2f} ms")
mq = max(byq, key=lambda q: sum(e - s for _, s, e, _ in byq[q]))
rs = byq[mq]
gaps = defaultdict(lambda: [0, 0.0])
tot = 0
for a, b in zip(rs, rs[1:]):
    g = b[1] - a[2]
    if g > 0:
        k = (a[0].split("(")[0][:44], b[0].split("(")[0][:44])
        gaps[k][0] += 1
        gaps[k][1] += g / 1e3
        tot += g
print(f"queue {mq} idle between its kernels: {tot / 1e6:.2f} ms; by (before -> after), top 15:")
for k, (c, us) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:15]:
    print(f"    {c:4d} x {us / c:6.1f} us = {us:7.1f} us  {k[0]} -> {k[1]}")
# the long gaps one by one: when in the step, and what the other queues ran meanwhile
big = sorted(((b[1] - a[2], a, b) for a, b in zip(rs, rs[1:]) if b[1] - a[2] > 20000), key=lambda t: -t[0])[:10]
for g, a, b in sorted(big, key=lambda t: t[1][2]):
    others = defaultdict(float)
    for n, s, e, q in step:
        if q != mq and e > a[2] and s < b[1]:
            others[(q, n.split("(")[0][:40])] += (min(e, b[1]) - max(s, a[2])) / 1e3
    txt = ", ".join(f"q{q} {n} {us:.0f} us" for (q, n), us in sorted(others.items(), key=lambda kv: -kv[1])[:3])
    print(f"    gap of {g / 1e3:6.1f} us at {(a[2] - t0) / 1e6:6.2f} ms: {a[0].split('(')[0][:40]} -> {b[0].split('(')[0][:40]} | meanwhile: {txt}")
# every kernel (all queues) that ran during the two longest gaps or ended / started within 150 us of them
for g, a, b in sorted(big, key=lambda t: -t[0])[:2]:
    print(f"--- around the gap of {g / 1e3:.1f} us at {(a[2] - t0) / 1e6:.2f} ms (start / end in us relative to the gap's begin):")
    for n, s_, e, q in step:
        if e > a[2] - 150000 and s_ < b[1] + 150000:
            print(f"      q{q} {(s_ - a[2]) / 1e3:9.1f} .. {(e - a[2]) / 1e3:9.1f}  {n.split('(')[0][:60]}")
for q, rs in sorted(byq.items()):
    agg = defaultdict(lambda: [0, 0.0])
    for n, s, e, _ in rs:
        k = n.split("(")[0][:70]
        agg[k][0] += 1
        agg[k][1] += (e - s) / 1e3
    print(f"--- queue {q}: per-kernel totals, top 22")
    for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
        print(f"    {c:4d} x {us / c:7.1f} us = {us:8.1f} us  {k}")
