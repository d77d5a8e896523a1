"""Eager vs recorded step (TrainStep(replay=True)): ATen audit of the recorded region, loss / grad-norm trajectories from the same seed,
ms per step and host enqueue time (dev tooling)."""
import os
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from cinema_amd import CineMA  # noqa: E402
from cinema_amd.optim import TrainStep  # noqa: E402

B = int(os.environ.get("BATCH", "16"))
kw = bench.base_kwargs("base")
batches = [bench.synthetic_batch(kw, B, i, "cuda") for i in range(2)]
traj = {}
for mode in ("eager", "replay", "eager", "replay"):
    torch.manual_seed(0)
    model = CineMA(**kw).to("cuda")
    step = TrainStep(model, replay=(mode == "replay"), audit=(mode == "replay" and "audited" not in traj))
    out = []
    for i in range(24):
        l, gn, m = step(batches[i % 2], 0.75)
        if i < 6 or i == 23:
            out.append((round(float(l), 6), round(float(gn), 5)))
    if mode == "replay" and "audited" not in traj:
        rec = next(iter(step._recorded.values()))
        traj["audited"] = True
        print(f"recorded {rec.n_launches} launches + {len(rec.calls) - rec.n_launches} host entries; unaccounted ATen ops in the region: {len(rec.unaccounted)}")
        from collections import Counter
        for k, n in Counter(rec.unaccounted).most_common(40):
            print(f"   {n:4d} x {k}")
    torch.cuda.synchronize()
    K_ = 40
    c0, t0 = os.times(), time.perf_counter()
    for i in range(K_):
        step(batches[i % 2], 0.75)
    t1, c1 = time.perf_counter(), os.times()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    cpu = (c1.user - c0.user + c1.system - c0.system) / K_ * 1e3
    print(f"{mode}: {1e3 * (t2 - t0) / K_:.2f} ms/step (enqueue {1e3 * (t1 - t0) / K_:.2f}, process CPU {cpu:.1f}); trajectory {out}", flush=True)
    del step, model
