"""Kernel sequence of the stem phases of the last step in a rocprofv3 kernel trace (dev tooling): per kernel name calls / total / mean, the idle time
between consecutive kernels, and the first N kernels in order.  Usage: phase_dump.py db [fwd|bwd] [n_list]"""
import sqlite3
import sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
which = sys.argv[2] if len(sys.argv) > 2 else "bwd"
n_list = int(sys.argv[3]) if len(sys.argv) > 3 else 120
rows = con.execute("select name, start, end from kernels order by start").fetchall()
rows = [(n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60], s, e) for n, s, e in rows]
adam = [i for i, r in enumerate(rows) if r[0].startswith("adamw_kernel")]
ends = [i for j, i in enumerate(adam) if j + 1 == len(adam) or adam[j + 1] - i > 8]
step = rows[ends[-2] + 1:ends[-1] + 1]
if which == "bwd":
    a = max(i for i, r in enumerate(step) if r[0].startswith(("attn_bwd_dkv_mfma<64>", "attn_bwd_dq_mfma<64>"))) + 12
    b = next(i for i, r in enumerate(step) if r[0].startswith("sqnorm"))
else:
    a, b = 0, next(i for i, r in enumerate(step) if r[0].startswith("attn_fwd_mfma<64>")) - 3
seg = step[a:b]
agg = defaultdict(lambda: [0, 0.0])
for n, s, e in seg:
    agg[n][0] += 1
    agg[n][1] += (e - s) / 1e3
print(f"{which}: {len(seg)} kernels, wall {(seg[-1][2] - seg[0][1]) / 1e6:.2f} ms, kernel time {sum(v[1] for v in agg.values()) / 1e3:.2f} ms")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"  {c:4d} x {t / c:7.1f} us = {t / 1e3:6.3f} ms  {n}")
print("in order (start offset us, duration us, gap to previous end us):")
t0, prev = seg[0][1], seg[0][1]
for n, s, e in seg[:n_list]:
    print(f"  {(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {(s - prev) / 1e3:7.1f}  {n}")
    prev = max(prev, e)
