"""K-sweep / epilogue-variant timing of the forward GEMM (dev tooling)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
m, n = 10960, 3072
for variant in (0, 3):
    print(f"variant {variant} (0=auto/big 256x128 3-stage, 3=128x128 2-stage glds)")
    for k in (64, 128, 256, 512, 768, 1536, 3072, 6144):
        x = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        bias = torch.randn(n, device=dev)
        y16 = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        y32 = torch.empty(m, n, dtype=torch.float32, device=dev)
        t1 = timeit(lambda: K.gemm(x, w, bias=bias, out=y16, force_generic=variant))
        t2 = timeit(lambda: K.gemm(x, w, out=y16, force_generic=variant))
        t3 = timeit(lambda: K.gemm(x, w, out=y32, force_generic=variant))
        fl = 2.0 * m * n * k
        print(f"  K={k:5d}: bf16+bias {t1*1e6:7.1f} us ({fl/t1/1e12:6.1f} TF) | bf16 {t2*1e6:7.1f} us | f32 {t3*1e6:7.1f} us", flush=True)
