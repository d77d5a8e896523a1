"""Which kernels surround the runtime's blit-copy kernels (__amd_rocclr_copyBuffer / fillBuffer) in a rocprofv3 kernel trace, in launch order
(dev tooling): python tools/copy_neighbors.py db [pattern]"""
import sqlite3
import sys
from collections import Counter

con = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "rocclr"
rows = con.execute("select name, start, end from kernels order by start").fetchall()


def short(n: str) -> str:
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]


cnt = Counter()
for i, (n, s, e) in enumerate(rows):
    if pat in n:
        prev = short(rows[i - 1][0]) if i else "-"
        nxt = short(rows[i + 1][0]) if i + 1 < len(rows) else "-"
        cnt[(prev, short(n), nxt, round((e - s) / 1e3))] += 1
for (p, n, x, us), c in cnt.most_common(40):
    print(f"{c:5d}  {p:60s} -> {n} ({us} us) -> {x}")
