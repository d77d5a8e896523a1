"""Weight-gradient layout (both operands reduction-strided): 256x128 wave-specialised kernel vs 128x128 kernel over split-K factors (dev tooling)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
for n_out, k_out, rows in ((768, 3072, 10960), (3072, 768, 10960), (2304, 768, 10960), (512, 2048, 32848), (2048, 512, 32848), (512, 512, 32848), (768, 768, 10960)):
    dy = (torch.randn(rows, n_out, device=dev) * 0.5).to(torch.bfloat16)
    x = (torch.randn(rows, k_out, device=dev) * 0.5).to(torch.bfloat16)
    dw = torch.zeros(n_out, k_out, device=dev)
    fl = 2.0 * rows * n_out * k_out
    line = [f"dW[{n_out}x{k_out}] over {rows} rows:"]
    for kind, fg in (("ws", 4), ("128", 3)):
        best = None
        for sk in (1, 2, 3, 4, 6, 8, 12, 16, 32):
            t = timeit(lambda: K.gemm(dy, x, a_kmajor=False, b_kmajor=False, out=dw, accumulate=True, split_k=sk, force_generic=fg))
            line.append(f"{kind}/sk{sk} {t * 1e6:.0f}")
            if best is None or t < best[0]:
                best = (t, sk)
        line.append(f"=> {kind} best sk{best[1]} {best[0] * 1e6:.1f} us ({fl / best[0] / 1e12:.0f} TF) |")
    print(" ".join(line), flush=True)
