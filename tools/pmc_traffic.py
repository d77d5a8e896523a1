"""Per-kernel mean HBM traffic per launch from two rocprofv3 PMC databases (FETCH_SIZE pass, WRITE_SIZE pass).

Corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: both counters are in KiB; on gfx950 FETCH_SIZE tallies the
128-B requests of wide coalesced reads at 64 B, so the read side is doubled (all our streaming reads are 16 B/lane).
usage: python tools/pmc_traffic.py fetch.db write.db > pmc_traffic.json
"""
import json
import sqlite3
import sys
from collections import defaultdict


def per_kernel(db: str, counter: str) -> dict:
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    ik = cols.index("kernel_name")
    ic = cols.index("counter_name") if "counter_name" in cols else cols.index("name")
    iv = cols.index("value") if "value" in cols else cols.index("counter_value")
    idd = cols.index("dispatch_id") if "dispatch_id" in cols else None
    per_dispatch = defaultdict(float)  # one row per (dispatch, XCD/instance): sum them
    names = {}
    for n, r in enumerate(con.execute("select * from counters_collection")):
        if r[ic] != counter:
            continue
        key = r[idd] if idd is not None else n
        per_dispatch[key] += float(r[iv])
        names[key] = r[ik]
    agg = defaultdict(lambda: [0.0, 0])
    for key, v in per_dispatch.items():
        a = agg[names[key]]
        a[0] += v
        a[1] += 1
    return {k: (s / n, n) for k, (s, n) in agg.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, (0, 0))[0] * fetch.get(k, (0, 0))[1])):
    f, nf = fetch.get(k, (0.0, 0))
    w, nw = write.get(k, (0.0, 0))
    out[k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]] = {"launches": nf or nw, "fetch_kib_raw": round(f, 1), "write_kib_raw": round(w, 1),
                                 "hbm_bytes_per_launch": round((2.0 * f + w) * 1024.0)}
json.dump({"note": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch (gfx950 FETCH_SIZE half-count correction)", "kernels": out}, sys.stdout, indent=1)
