"""Achievable HBM write / copy bandwidth on this box (torch fill_/copy_ and the library's cast kernel), for the GEMM epilogue budget (dev tooling)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for mb in (32, 134, 268, 1024):
    n = mb * 1000 * 1000 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    y = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    f32 = torch.empty(n, dtype=torch.float32, device="cuda")
    t = timeit(lambda: x.fill_(1.0))
    t2 = timeit(lambda: y.copy_(x))
    t3 = timeit(lambda: K.cast(f32, torch.bfloat16, out=y))
    print(f"{mb:5d} MB: fill {mb / 1e3 / t / 1e3:.2f} TB/s ({t * 1e6:.0f} us) | copy {2 * mb / 1e3 / t2 / 1e3:.2f} TB/s r+w ({t2 * 1e6:.0f} us) | "
          f"cast f32->bf16 {3 * mb / 1e3 / t3 / 1e3:.2f} TB/s r+w ({t3 * 1e6:.0f} us)")
