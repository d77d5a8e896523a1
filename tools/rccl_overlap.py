"""Do the gradient collectives run beside the backward kernels?  From a rocprofv3 kernel trace of `bench.py --force-sync` (one-process RCCL group
on one GPU, or a real multi-GPU run): for the last step, every RCCL kernel with its start / end inside the step and the compute kernels whose
execution intervals overlap it.  Usage: python tools/rccl_overlap.py results.db > profiles/r02_force_sync_timeline.txt"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
rows = [(n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70], s, e) for n, s, e in rows]
adam = [i for i, r in enumerate(rows) if r[0].startswith("adamw_kernel")]
ends = [i for j, i in enumerate(adam) if j + 1 == len(adam) or adam[j + 1] - i > 8]
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]
t0, t1 = step[0][1], step[-1][2]
is_coll = lambda n: "nccl" in n.lower() or "rccl" in n.lower()  # noqa: E731
coll = [r for r in step if is_coll(r[0])]
comp = [r for r in step if not is_coll(r[0])]
first_bwd = next((r[1] for r in step if r[0].startswith("attn_bwd")), t0)
print(f"last step: {len(step)} kernels, {(t1 - t0) / 1e6:.2f} ms wall; backward starts at +{(first_bwd - t0) / 1e6:.2f} ms; {len(coll)} collective kernels")
print("  start_ms    end_ms   dur_us  overlapped_compute_kernels  overlapped_compute_us  name")
tot = ov_tot = 0.0
for n, s, e in coll:
    ov = [(max(s, cs), min(e, ce)) for _, cs, ce in comp if cs < e and ce > s]
    ov_us = sum(b - a for a, b in ov) / 1e3
    tot += (e - s) / 1e3
    ov_tot += min(ov_us, (e - s) / 1e3)
    print(f"  {(s - t0) / 1e6:8.3f}  {(e - t0) / 1e6:8.3f}  {(e - s) / 1e3:7.1f}  {len(ov):6d}  {ov_us:10.1f}  {n}")
if coll:
    print(f"collective kernel time {tot / 1e3:.2f} ms, of which {ov_tot / 1e3:.2f} ms ran while a compute kernel was executing "
          f"({100 * ov_tot / max(tot, 1e-9):.0f} %); last collective ends at +{(coll[-1][2] - t0) / 1e6:.2f} ms, the optimiser (sqnorm) starts at "
          f"+{(next(r[1] for r in step if r[0].startswith('sqnorm')) - t0) / 1e6:.2f} ms")
