"""256x128 wave-specialised kernel vs 128x128 kernel on forward-layout (K-major) GEMMs with a plain fp32 output (dev tooling)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
for m, n, k in ((4096, 4096, 4096), (8192, 8192, 8192), (10960, 768, 3072), (32848, 512, 2048), (10960, 3072, 768), (10960, 2304, 768)):
    x = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    y32 = torch.empty(m, n, dtype=torch.float32, device=dev)
    t128 = timeit(lambda: K.gemm(x, w, out=y32, force_generic=3))
    tws = timeit(lambda: K.gemm(x, w, out=y32, force_generic=4))
    fl = 2.0 * m * n * k
    print(f"{m}x{n}x{k}: 128x128 {t128*1e6:8.1f} us ({fl/t128/1e12:6.1f} TF) | ws 256x128 {tws*1e6:8.1f} us ({fl/tws/1e12:6.1f} TF)", flush=True)
