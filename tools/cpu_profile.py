"""Host-side cost of one step: cProfile of the forward/optimiser thread and of the autograd (backward) thread, by own time (dev tooling)."""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from cinema_amd import CineMA  # noqa: E402
from cinema_amd import tape as T  # noqa: E402
from cinema_amd.optim import TrainStep  # noqa: E402

kw = bench.base_kwargs("base")
torch.manual_seed(0)
model = CineMA(**kw).to("cuda")
step = TrainStep(model)
import os
B = int(os.environ.get("BATCH", "16"))
batch = bench.synthetic_batch(kw, B, 1, "cuda")
for _ in range(10):
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

bw_prof = cProfile.Profile()
orig = T._TapedCall.backward


def profiled_backward(ctx, *grads):  # noqa: ANN001, ANN002, ANN201
    bw_prof.enable()
    try:
        return orig(ctx, *grads)
    finally:
        bw_prof.disable()


T._TapedCall.backward = staticmethod(profiled_backward)
fw_prof = cProfile.Profile()
N = 5
fw_prof.enable()
for _ in range(N):
    step(batch, 0.75)
fw_prof.disable()
torch.cuda.synchronize()
for name, pr in (("forward + optimiser thread", fw_prof), ("backward thread", bw_prof)):
    print(f"==== {name} ({N} steps) ====")
    pstats.Stats(pr).sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 30)
