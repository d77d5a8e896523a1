"""The end of the backward pass in a rocprofv3 kernel trace (rocpd sqlite): every kernel that runs in the last `window` ms before the optimiser's first kernel of the last
step, with its queue (= HIP stream), start offset and duration (dev tooling: what the main stream waits for at the final join).
   tail_timeline.py results.db [window_ms]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
rows = con.execute("select name, start, end, queue_id from kernels order by start").fetchall()
rows = [(n.replace("(anonymous namespace)::", "").replace("void ", ""), s, e, q) for n, s, e, q in rows]
sq = [i for i, r in enumerate(rows) if r[0].startswith("sqnorm_kernel")]
i_end = sq[-1]
t_end = rows[i_end][1]
sel = [r for r in rows[:i_end] if r[2] > t_end - win * 1e6]
qs = sorted({r[3] for r in sel})
print(f"last {win} ms before sqnorm: {len(sel)} kernels on queues {qs}")
for n, s, e, q in sel:
    print(f"  q{qs.index(q)}  {(s - t_end) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {n[:90]}")
