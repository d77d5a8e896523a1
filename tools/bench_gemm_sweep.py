"""K-sweep timing of the forward GEMM for the encoder/decoder tile grids (dev tooling): time = intercept + slope * K."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
for m, n in ((10960, 768), (10960, 3072), (32848, 512), (10752, 768), (8192, 1024)):
    tiles = ((m + 127) // 128) * ((n + 127) // 128)
    print(f"M={m} N={n}: {tiles} tiles = {tiles / 512:.2f} rounds of 512")
    for k in (64, 256, 768, 1536, 3072, 6144):
        x = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        y16 = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        t2 = timeit(lambda: K.gemm(x, w, out=y16))
        fl = 2.0 * m * n * k
        print(f"  K={k:5d}: {t2*1e6:7.1f} us ({fl/t2/1e12:6.1f} TF)", flush=True)
