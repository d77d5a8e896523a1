"""The stem's thin GEMMs (large M, N and K of 64-512): time, TF and effective HBM rate against the byte floor (dev tooling)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
for m, n, k in ((147456, 64, 64), (147456, 256, 64), (147456, 64, 256), (36864, 128, 128), (36864, 512, 128), (36864, 128, 512), (9216, 64, 64), (9216, 256, 64), (2304, 128, 128)):
    x = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(n, device=dev)
    y16 = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    y32 = torch.empty(m, n, dtype=torch.float32, device=dev)
    res = torch.randn(m, n, device=dev)
    t1 = timeit(lambda: K.gemm(x, w, out=y16, bias=bias))
    t2 = timeit(lambda: K.gemm(x, w, out=y32, bias=bias, residual=res))
    dy = (torch.randn(m, n, device=dev) * 0.5).to(torch.bfloat16)
    dx = torch.empty(m, k, dtype=torch.bfloat16, device=dev)
    t3 = timeit(lambda: K.gemm(dy, w, a_kmajor=True, b_kmajor=False, out=dx))
    b1 = (m * k + n * k + m * n) * 2
    b2 = (m * k + n * k) * 2 + m * n * 8
    tiles = ((m + 127) // 128) * ((n + 127) // 128)
    print(f"{m}x{n}x{k} ({tiles} tiles): bf16 out {t1 * 1e6:6.1f} us ({b1 / t1 / 1e12:4.2f} TB/s of the byte floor) | fp32+res {t2 * 1e6:6.1f} us ({b2 / t2 / 1e12:4.2f} TB/s) | dgrad {t3 * 1e6:6.1f} us", flush=True)
