"""Practical bf16 GEMM ceiling on this box: torch.mm (hipBLASLt / rocBLAS) next to this library's kernel on the same shapes (dev tooling;
the library itself never calls a BLAS)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
for m, n, k in ((8192, 8192, 8192), (32848, 2048, 512), (10960, 3072, 768), (10960, 768, 3072), (32848, 512, 2048)):
    x = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    y = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    t_blas = timeit(lambda: torch.mm(x, w.t(), out=y))
    t_ours = timeit(lambda: K.gemm(x, w, out=y))
    fl = 2.0 * m * n * k
    print(f"{m}x{n}x{k}: torch.mm {t_blas * 1e6:8.1f} us ({fl / t_blas / 1e12:6.0f} TF) | cinema_gemm_bf16 {t_ours * 1e6:8.1f} us ({fl / t_ours / 1e12:6.0f} TF)", flush=True)
