"""Cost of the 8-bit output copy in the e4m3 GEMM epilogues at the ViT-Large MLP shapes (dev tooling)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402
from cinema_amd import tape as T  # noqa: E402
from tools.bench_p256 import bench  # noqa: E402

dev = "cuda"
m, n, k = 13824, 4096, 1024
x = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
bias = torch.randn(n, device=dev)
x8, sx = K.quantize_fp8(x)
w8, sw = K.quantize_fp8(w)
h = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
sites = T.Fp8Sites(torch.device(dev, 0))
s = sites.site(("a", 0))
K.gemm_fp8(x8, sx, w8, sw, bias=bias, act=1, aux_out=h, gelu_deriv=True, out8=(s, None))
sites.update()
o8 = torch.empty(m, n, dtype=torch.uint8, device=dev)
strips = torch.empty((m + 31) // 32, n, device=dev)
gin = (torch.randn(m, n, device=dev)).to(torch.bfloat16)
fns = {
    "gelu plain": lambda: K.gemm_fp8(x8, sx, w8, sw, bias=bias, act=1, aux_out=h, gelu_deriv=True),
    "gelu amax only": lambda: K.gemm_fp8(x8, sx, w8, sw, bias=bias, act=1, aux_out=h, gelu_deriv=True, out8=(s, None)),
    "gelu + out8": lambda: K.gemm_fp8(x8, sx, w8, sw, bias=bias, act=1, aux_out=h, gelu_deriv=True, out8=(s, o8)),
    "gelu out8 only": lambda: K.gemm_fp8(x8, sx, w8, sw, bias=bias, act=1, aux_out=h, gelu_deriv=True, out8=(s, o8), skip_d=True),
    "gelu no aux": lambda: K.gemm_fp8(x8, sx, w8, sw, bias=bias, act=1),
    "plain bf16 out": lambda: K.gemm_fp8(x8, sx, w8, sw, bias=bias),
    "x gelu' plain": lambda: K.gemm_fp8(x8, sx, w8, sw, gelu_in=gin, gelu_deriv=True),
    "x gelu' + out8 + strips": lambda: K.gemm_fp8(x8, sx, w8, sw, gelu_in=gin, gelu_deriv=True, out8=(s, o8), colsum_partials=strips),
    "x gelu' out8 only + strips": lambda: K.gemm_fp8(x8, sx, w8, sw, gelu_in=gin, gelu_deriv=True, out8=(s, o8), colsum_partials=strips, skip_d=True),
}
r = bench(fns, iters=6)
for name, t in r.items():
    print(f"{name:28s} {t * 1e6:7.1f} us  {2.0 * m * n * k / t / 1e12:6.0f} TF", flush=True)
