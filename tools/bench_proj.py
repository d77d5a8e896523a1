"""The residual-stream projections (fp32 output + bias + fp32 residual; proj / fc2 forward) at whole-round and at the step's ragged row counts (dev tooling).
   python tools/bench_proj.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
for name, m, n, k in (("enc proj whole", 10880, 768, 768), ("enc proj step", 10960, 768, 768), ("enc proj +1 strip", 11008, 768, 768),
                      ("enc fc2 whole", 10880, 768, 3072), ("enc fc2 step", 10960, 768, 3072),
                      ("dec proj whole", 32768, 512, 512), ("dec proj step", 32848, 512, 512),
                      ("dec fc2 whole", 32768, 512, 2048), ("dec fc2 step", 32848, 512, 2048)):
    x = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(n, device=dev)
    res = torch.randn(m, n, device=dev)
    y = torch.empty(m, n, dtype=torch.float32, device=dev)
    t = timeit(lambda: K.gemm(x, w, bias=bias, residual=res, out=y), iters=20)
    t16 = timeit(lambda: K.gemm(x, w, bias=bias), iters=20)
    nbytes = 2 * m * k + 2 * n * k + 8 * m * n
    print(f"{name:18s} {m:6d} {n:5d} {k:5d} | fp32+residual {t * 1e6:7.1f} us {2.0 * m * n * k / t / 1e12:6.0f} TF {nbytes / t / 1e12:5.2f} TB/s | bf16 out {t16 * 1e6:7.1f} us", flush=True)
