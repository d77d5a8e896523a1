"""K-sweep timing of the forward GEMM for the encoder/decoder tile grids (dev tooling): time = intercept + slope * K."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
shapes = ((10960, 768), (10960, 3072), (32848, 512), (32848, 2048), (10752, 768))
ks = (256, 512, 768, 1536, 3072)
for m, n in shapes:
    tiles = ((m + 127) // 128) * ((n + 127) // 128)
    print(f"M={m} N={n}: {tiles} tiles = {tiles / 512:.2f} rounds of 512")
    for k in ks:
        x = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        bias = torch.randn(n, device=dev)
        y16 = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        pre = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        t2 = timeit(lambda: K.gemm(x, w, out=y16))
        t3 = timeit(lambda: K.gemm(x, w, out=y16, bias=bias, act=1, aux_out=pre))
        fl = 2.0 * m * n * k
        print(f"  K={k:5d}: plain {t2*1e6:7.1f} us ({fl/t2/1e12:6.1f} TF) | bias+gelu+preact {t3*1e6:7.1f} us ({fl/t3/1e12:6.1f} TF)", flush=True)
