"""Per-kernel MFMA utilisation from rocprofv3 PMC databases (any number of passes; counters are merged per kernel name).

For every dispatch the per-instance rows of a counter (one per XCD / SE) are summed; GRBM_GUI_ACTIVE is averaged over its instances instead
(it is the same wall-clock interval seen by every XCD).  Per kernel the dispatches are averaged.

  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (gpu_cycles * 256 CUs * 4 SIMDs)   (busy cycles are per SIMD, summed over the chip;
              gpu_cycles = GRBM_GUI_ACTIVE / 8 XCDs: rocprofv3 reports the sum over the XCDs' GRBMs as one row)
  cross-check printed per kernel: implied_clock_ghz = gpu_cycles / kernel-trace duration must be a plausible shader clock (1.8-2.4)

usage: python tools/pmc_mfma.py pass1.db [pass2.db ...] > r02_mfma_util.json
"""
import json
import sqlite3
import sys
from collections import defaultdict

N_SIMD = 256 * 4
N_XCD = 8


def short(name: str) -> str:
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:100]


def load(db: str) -> dict:
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    ik = cols.index("kernel_name")
    ic = cols.index("counter_name") if "counter_name" in cols else cols.index("name")
    iv = cols.index("value") if "value" in cols else cols.index("counter_value")
    idd = cols.index("dispatch_id") if "dispatch_id" in cols else None
    per = defaultdict(lambda: [0.0, 0])  # (dispatch, counter) -> [sum, rows]
    names = {}
    for n, r in enumerate(con.execute("select * from counters_collection")):
        key = (r[idd] if idd is not None else n, r[ic])
        per[key][0] += float(r[iv])
        per[key][1] += 1
        names[key[0]] = r[ik]
    out = defaultdict(lambda: defaultdict(lambda: [0.0, 0, 0]))  # kernel -> counter -> [sum over dispatches, dispatches, rows per dispatch]
    for (disp, ctr), (s, rows) in per.items():
        a = out[short(names[disp])][ctr]
        a[0] += s
        a[1] += 1
        a[2] = rows
    # kernel durations from the kernel trace of the same pass (ns), when the view exists
    try:
        kcols = [r[1] for r in con.execute("pragma table_info(kernels)")]
        if {"name", "start", "end"} <= set(kcols):
            for name, st, en in con.execute("select name, start, end from kernels"):
                a = out[short(name)]["_duration_ns"]
                a[0] += float(en - st)
                a[1] += 1
                a[2] = 1
    except sqlite3.Error:
        pass
    return out


merged: dict = defaultdict(dict)
for db in sys.argv[1:]:
    for k, ctrs in load(db).items():
        for c, (s, n, rows) in ctrs.items():
            if c == "_duration_ns" and c in merged[k]:
                continue
            merged[k][c] = {"mean": s / n, "launches": n, "rows": rows}

table = {}
for k, ctrs in merged.items():
    if not any(t in k for t in ("gemm", "attn", "mfma")):
        continue
    row = {c: round(v["mean"], 1) for c, v in ctrs.items() if not c.startswith("_")}
    row["launches"] = max(v["launches"] for v in ctrs.values())
    busy = ctrs.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("mean")
    gui = ctrs.get("GRBM_GUI_ACTIVE", {}).get("mean")
    if gui:  # one row per dispatch = the sum over the 8 XCDs (each XCD's GRBM counts the same wall-clock interval); more rows: already per instance
        gui = gui / (N_XCD if ctrs["GRBM_GUI_ACTIVE"]["rows"] == 1 else ctrs["GRBM_GUI_ACTIVE"]["rows"])
        row["gpu_cycles"] = round(gui, 1)
        row["rows_per_dispatch"] = {c: v["rows"] for c, v in ctrs.items() if not c.startswith("_")}
    dur = ctrs.get("_duration_ns", {}).get("mean")
    if dur:
        row["duration_us_profiled"] = round(dur / 1e3, 2)
        if gui:
            row["implied_clock_ghz"] = round(gui / dur, 3)
    if busy is not None and gui:
        row["mfma_util"] = round(busy / (gui * N_SIMD), 4)
    if busy is not None and dur:  # against the kernel-trace duration at the 2.4 GHz peak clock (= achieved / peak FLOP rate of the MFMA work)
        row["mfma_util_wall_2p4ghz"] = round(busy / (dur * 2.4 * N_SIMD), 4)
    sqb = ctrs.get("SQ_BUSY_CYCLES", {}).get("mean")
    if busy is not None and sqb:
        row["mfma_busy_over_sq_busy"] = round(busy / sqb, 4)
    table[k] = row
order = sorted(table, key=lambda k: -(table[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) * table[k]["launches"]))
json.dump({"note": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES (summed over all SIMDs of the chip) / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), per-kernel mean over "
                   "launches; counters from separate rocprofv3 --pmc passes of `bench.py --steps 2` (one stream)",
           "kernels": {k: table[k] for k in order}}, sys.stdout, indent=1)
