"""Phase timeline of the last step in a rocprofv3 kernel trace (rocpd sqlite): where the wall time of one step goes, and how busy the GPU is
inside each phase.  Usage: phase_timeline.py mae_results.db"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
kd = next(t for t in tabs if t.startswith("kernels") or t == "kernels")
cols = [r[1] for r in con.execute(f"pragma table_info({kd})")]
rows = con.execute(f"select name, start, end from {kd} order by start").fetchall()
rows = [(n.replace("(anonymous namespace)::", "").replace("void ", ""), s, e) for n, s, e in rows]
adam = [i for i, r in enumerate(rows) if r[0].startswith("adamw")]
ends = [i for j, i in enumerate(adam) if j + 1 == len(adam) or adam[j + 1] - i > 8]  # last AdamW launch of every step (one per param group)
lo, hi = ends[-2] + 1, ends[-1] + 1  # the last full step
step = rows[lo:hi]
t0 = step[0][1]


def first(pred, begin=0):
    return next(i for i in range(begin, len(step)) if pred(step[i][0]))


def last(pred):
    return max(i for i in range(len(step)) if pred(step[i][0]))


marks = [("stems fwd (4 views) + token assembly", 0)]
i_enc = first(lambda n: n.startswith("attn_fwd_mfma<64"))
marks.append(("encoder fwd", i_enc - 3))
i_dec = first(lambda n: n.startswith("attn_fwd_mfma<32"))
marks.append(("fusion + decoder fwd + loss", last(lambda n: n.startswith("attn_fwd_mfma<64")) + 8))
i_bwd = first(lambda n: n.startswith("attn_bwd") or n.startswith("attn_delta"))
marks.append(("decoder bwd", i_bwd - 12))
marks.append(("fusion bwd + encoder bwd", last(lambda n: n.startswith("attn_bwd_dkv_mfma<32>") or n.startswith("attn_bwd_dq_mfma<32>") or n.startswith("attn_bwd_fused_mfma<32>")) + 12))
marks.append(("stems bwd", last(lambda n: n.startswith("attn_bwd_dkv_mfma<64>") or n.startswith("attn_bwd_dq_mfma<64>")) + 12))
marks.append(("clip + AdamW", first(lambda n: n.startswith("sqnorm_kernel"))))
marks.append(("end", len(step)))
print(f"step: {len(step)} kernels, {(step[-1][2] - t0) / 1e6:.2f} ms wall")
for (name, a), (_, b) in zip(marks, marks[1:]):
    seg = step[a:b]
    if not seg:
        continue
    wall = (seg[-1][2] - seg[0][1]) / 1e6
    # union of busy intervals (two streams overlap)
    busy, cur_s, cur_e = 0, None, None
    for _, s, e in sorted(seg, key=lambda r: r[1]):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"  {name:40s} {len(seg):5d} kernels  wall {wall:6.2f} ms  GPU busy {busy / 1e6:6.2f} ms ({100 * busy / 1e6 / wall:3.0f}%)  avg kernel {sum(e - s for _, s, e in seg) / len(seg) / 1e3:6.1f} us")

import os  # noqa: E402

dump = os.environ.get("PHASE_DUMP")  # e.g. "stems": per-kernel totals of every phase whose name contains the string
if dump:
    for (name, a), (_, b) in zip(marks, marks[1:]):
        if dump not in name:
            continue
        agg: dict = {}
        for n, s, e in step[a:b]:
            k = n.split("(")[0][:70]
            t = agg.setdefault(k, [0, 0.0])
            t[0] += 1
            t[1] += (e - s) / 1e3
        print(f"--- {name}")
        for k, (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
            print(f"    {cnt:4d} x {us / cnt:7.1f} us = {us:8.1f} us  {k}")

# idle gaps of the whole step (no kernel of either stream running): total, and the largest ones with the kernels on both sides
ev = sorted(step, key=lambda r: r[1])
gaps, cur_e, cur_n = [], ev[0][2], ev[0][0]
for n, s, e in ev[1:]:
    if s > cur_e:
        gaps.append((s - cur_e, cur_n, n, (cur_e - t0) / 1e6))
    if e > cur_e:
        cur_e, cur_n = e, n
tot = sum(g[0] for g in gaps)
print(f"idle (no kernel running): {tot / 1e6:.2f} ms in {len(gaps)} gaps; gaps > 5 us: {sum(1 for g in gaps if g[0] > 5e3)} = {sum(g[0] for g in gaps if g[0] > 5e3) / 1e6:.2f} ms")
hist: dict = {}
for g in gaps:
    k = (g[1].split("(")[0].split("<")[0][:40], g[2].split("(")[0].split("<")[0][:40])
    t = hist.setdefault(k, [0, 0.0])
    t[0] += 1
    t[1] += g[0] / 1e3
print("idle by (kernel before -> kernel after), top 25:")
for k, (cnt, us) in sorted(hist.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"    {cnt:4d} x {us / cnt:6.1f} us = {us:8.1f} us  {k[0]} -> {k[1]}")
