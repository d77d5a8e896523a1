"""Is the step launch-bound?  CPU time to enqueue K steps vs time until the GPU has finished them (dev tooling)."""
import sys
import time
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
for _ in range(3):
    step(batch, 0.75)
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K):
    step(batch, 0.75)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / K:.2f} ms/step; until GPU done {1e3 * (t2 - t0) / K:.2f} ms/step")
# forward / backward / optimiser split of the CPU side
import cProfile
import pstats

pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step(batch, 0.75)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
