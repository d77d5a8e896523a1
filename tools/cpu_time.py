"""Host CPU seconds consumed per step (all threads, os.times) next to the wall time per step: is the step launch-bound? (dev tooling)"""
import os
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from cinema_amd import CineMA  # noqa: E402
from cinema_amd.optim import TrainStep  # noqa: E402

kw = bench.base_kwargs("base")
for b in (16, 4, 1):
    torch.manual_seed(0)
    model = CineMA(**kw).to("cuda")
    step = TrainStep(model)
    batch = bench.synthetic_batch(kw, b, 1, "cuda")
    for _ in range(15):
        step(batch, 0.75)
    torch.cuda.synchronize()
    K = 30
    c0, t0 = os.times(), time.perf_counter()
    for _ in range(K):
        step(batch, 0.75)
    c1, t1 = os.times(), time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    cpu = (c1.user - c0.user + c1.system - c0.system) / K * 1e3
    print(f"batch {b:2d}: wall {1e3 * (t2 - t0) / K:.2f} ms/step, enqueue returns after {1e3 * (t1 - t0) / K:.2f} ms/step, process CPU time {cpu:.2f} ms/step "
          f"(user {1e3 * (c1.user - c0.user) / K:.2f} + sys {1e3 * (c1.system - c0.system) / K:.2f})", flush=True)
    del step, model
