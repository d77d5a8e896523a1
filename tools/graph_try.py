"""Experiment: capture forward+backward of the MAE step in a HIP graph (torch.cuda.graph) and replay it."""
import faulthandler
import sys
import time
faulthandler.enable()
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from cinema_amd import CineMA  # noqa: E402
from cinema_amd.optim import TrainStep  # noqa: E402

kw = bench.base_kwargs("base")
torch.manual_seed(0)
model = CineMA(**kw).to("cuda")
step = TrainStep(model)
batch = bench.synthetic_batch(kw, 16, 1, "cuda")
static = {k: v.clone() for k, v in batch.items()}
for _ in range(3):
    step(static, 0.75)
torch.cuda.synchronize()

g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
print('side-stream warmup', flush=True)
with torch.cuda.stream(s):
    for _ in range(2):  # warm-up on the side stream (allocator, per-stream workspaces)
        loss, _, _, _ = model(static, 0.75)
        loss.backward()
        step.optimizer.step(5.0)
        step.optimizer.zero_grad()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print('capturing', flush=True)
with torch.cuda.graph(g):
    loss, _, _, metrics = model(static, 0.75)
    loss.backward()
torch.cuda.synchronize()
print("captured")


def replay_step():
    g.replay()
    gn = step.optimizer.step(5.0)
    step.optimizer.zero_grad()
    return loss, gn


for _ in range(3):
    replay_step()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    l, gn = replay_step()
torch.cuda.synchronize()
print(f"graph replay: {1e3 * (time.perf_counter() - t0) / K:.2f} ms/step, loss {float(l):.5f} grad_norm {float(gn):.4f}")
t0 = time.perf_counter()
for _ in range(K):
    l2, gn2, _ = step(static, 0.75)
torch.cuda.synchronize()
print(f"eager: {1e3 * (time.perf_counter() - t0) / K:.2f} ms/step, loss {float(l2):.5f} grad_norm {float(gn2):.4f}")
