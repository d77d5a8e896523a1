"""Per-kernel mean of PMC counters from rocprofv3 rocpd databases."""
import sqlite3
import sys
from collections import defaultdict

for db in sys.argv[1:]:
    try:
        con = sqlite3.connect(db)
        cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
        rows = con.execute("select * from counters_collection").fetchall()
    except Exception as e:  # noqa: BLE001
        print(db, "ERR", e)
        continue
    ik, ic, iv = cols.index("kernel_name") if "kernel_name" in cols else None, None, None
    for cand in ("counter_name", "name"):
        if cand in cols:
            ic = cols.index(cand)
    for cand in ("value", "counter_value"):
        if cand in cols:
            iv = cols.index(cand)
    if ik is None or ic is None or iv is None:
        print(db, "columns:", cols)
        continue
    agg = defaultdict(lambda: [0.0, 0])
    for r in rows:
        a = agg[(r[ik][:70], r[ic])]
        a[0] += float(r[iv]); a[1] += 1
    print("==", db)
    for (k, c), (s, n) in sorted(agg.items()):
        if "gemm" in k or "attn" in k or "dwconv" in k:
            print(f"{k:70s} {c:28s} mean {s / n:16.1f}  (n={n})")
