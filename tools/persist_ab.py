"""In-process A/B: persistent vs one-shot 128x128 GEMM kernel (CINEMA_GEMM_ONESHOT is read at every launch)."""
import os
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from cinema_amd import CineMA  # noqa: E402
from cinema_amd import hip as K  # noqa: E402
from cinema_amd.optim import TrainStep  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
for m, n, k in ((10960, 3072, 768), (10960, 768, 3072), (32848, 2048, 512), (32848, 512, 2048), (10960, 2304, 768), (32848, 512, 512)):
    x = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(n, device=dev)
    y16 = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    res = {}
    for mode in ("1", "0", "1", "0"):
        os.environ["CINEMA_GEMM_PERSIST"] = "0" if mode == "1" else "1"
        res.setdefault(mode, []).append(timeit(lambda: K.gemm(x, w, out=y16, bias=bias)) * 1e6)
    print(f"{m}x{n}x{k}: one-shot {min(res['1']):.1f} us, persistent {min(res['0']):.1f} us")

kw = bench.base_kwargs("base")
torch.manual_seed(0)
model = CineMA(**kw).to(dev)
step = TrainStep(model)
batches = [bench.synthetic_batch(kw, 16, i, dev) for i in range(2)]


def run(n):
    for i in range(n):
        step(batches[i % 2], 0.75)
    torch.cuda.synchronize()


run(25)
for mode in ("1", "0", "1", "0", "1", "0"):
    os.environ["CINEMA_GEMM_PERSIST"] = "0" if mode == "1" else "1"
    run(4)
    t0 = time.perf_counter()
    run(30)
    print(f"oneshot={mode}: {1e3 * (time.perf_counter() - t0) / 30:.2f} ms/step", flush=True)
