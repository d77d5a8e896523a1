"""In-process A/B of the side-stream weight gradients (dev tooling)."""
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
batches = [bench.synthetic_batch(kw, 16, i, "cuda") for i in range(2)]


def run(n):
    for i in range(n):
        step(batches[i % 2], 0.75)
    torch.cuda.synchronize()


MODES = sys.argv[1:] or ["off", "on", "off", "on", "off", "on"]
run(8)
for mode in MODES:
    T.SIDE_WGRAD = mode != "off"
    run(4)
    t0 = time.perf_counter()
    run(30)
    print(f"{mode:8s} {1e3 * (time.perf_counter() - t0) / 30:.2f} ms/step  mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
