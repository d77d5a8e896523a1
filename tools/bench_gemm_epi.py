"""Cost of the fused epilogue pieces on the fc1-shaped forward GEMMs (dev tooling)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
for m, n, k in ((32848, 2048, 512), (10960, 3072, 768), (10960, 768, 3072), (32848, 512, 2048)):
    x = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(n, device=dev)
    y16 = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    y32 = torch.empty(m, n, dtype=torch.float32, device=dev)
    res = torch.randn(m, n, device=dev)
    pre = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    gin = (torch.randn(m, n, device=dev)).to(torch.bfloat16)
    cases = {
        "plain bf16": lambda: K.gemm(x, w, out=y16),
        "bias": lambda: K.gemm(x, w, out=y16, bias=bias),
        "bias+gelu": lambda: K.gemm(x, w, out=y16, bias=bias, act=1),
        "bias+preact": lambda: K.gemm(x, w, out=y16, bias=bias, aux_out=pre),
        "bias+gelu+preact": lambda: K.gemm(x, w, out=y16, bias=bias, act=1, aux_out=pre),
        "gelu_in (dgrad)": lambda: K.gemm(x, w, out=y16, gelu_in=gin),
        "f32 out": lambda: K.gemm(x, w, out=y32),
        "bias+res f32": lambda: K.gemm(x, w, out=y32, bias=bias, residual=res),
    }
    print(f"M={m} N={n} K={k}")
    for name, fn in cases.items():
        t = timeit(fn)
        print(f"  {name:18s} {t*1e6:7.1f} us ({2.0*m*n*k/t/1e12:6.1f} TF)", flush=True)
