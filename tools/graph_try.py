"""A/B: eager step vs forward+backward replayed as one HIP graph (TrainStep(hip_graph=True)); same seeds, prints loss trajectories."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from cinema_amd import CineMA  # noqa: E402
from cinema_amd.optim import TrainStep  # noqa: E402

kw = bench.base_kwargs("base")
batch = bench.synthetic_batch(kw, 16, 1, "cuda")
for mode in (False, True, False, True):
    torch.manual_seed(0)
    model = CineMA(**kw).to("cuda")
    step = TrainStep(model, hip_graph=mode)
    losses = []
    for i in range(25):
        l, gn, _ = step(batch, 0.75)
        if i < 4:
            losses.append((round(float(l), 5), round(float(gn), 4)))
    torch.cuda.synchronize()
    K = 40
    t0 = time.perf_counter()
    for _ in range(K):
        l, gn, _ = step(batch, 0.75)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"hip_graph={mode}: {1e3 * (t2 - t0) / K:.2f} ms/step (host enqueue {1e3 * (t1 - t0) / K:.2f}); final loss {float(l):.5f}; first steps {losses}", flush=True)
    del step, model
