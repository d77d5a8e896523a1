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
    step = TrainStep(model, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05, clip_grad=5.0, replay="--eager" not in sys.argv)
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
    # the same steps with the device drained before each: the host's own cost of issuing one step (no queue back-pressure), split replay / clip + AdamW
    h_rep, h_opt = 0.0, 0.0
    for _ in range(10):
        torch.cuda.synchronize()
        a = time.perf_counter()
        if step.replay:
            step._replay_step(batch, 0.75, False)
        b_ = time.perf_counter()
        step.optimizer.step(5.0)
        step.optimizer.zero_grad()
        c_ = time.perf_counter()
        h_rep += b_ - a
        h_opt += c_ - b_
    print(f"          host alone: replayed forward + backward issued in {1e2 * h_rep:.2f} ms, clip + AdamW + zero_grad issued in {1e2 * h_opt:.3f} ms", flush=True)
    cpu = (c1.user - c0.user + c1.system - c0.system) / K * 1e3
    print(f"batch {b:2d}: wall {1e3 * (t2 - t0) / K:.2f} ms/step, enqueue returns after {1e3 * (t1 - t0) / K:.2f} ms/step, process CPU time {cpu:.2f} ms/step "
          f"(user {1e3 * (c1.user - c0.user) / K:.2f} + sys {1e3 * (c1.system - c0.system) / K:.2f})", flush=True)
    del step, model
