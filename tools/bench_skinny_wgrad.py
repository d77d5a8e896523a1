"""Weight gradients with a small output and a very long reduction (the conv stem's 1x1 / k==s convs): split count sweep of the 128x128 kernel (dev tooling)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

SHAPES = [(64, 64, 147456), (128, 128, 36864), (256, 64, 147456), (64, 256, 147456), (128, 512, 36864), (512, 128, 36864), (768, 512, 9216), (64, 64, 9216)]
for n, k, rows in SHAPES:
    dy = (torch.randn(rows, n, device="cuda") * 0.5).to(torch.bfloat16)
    x = (torch.randn(rows, k, device="cuda") * 0.5).to(torch.bfloat16)
    dst = torch.zeros(n, k, dtype=torch.float32, device="cuda")
    rs = torch.zeros(n, dtype=torch.float32, device="cuda")
    res = []
    for sk in (16, 32, 64, 128, 256, 512):
        if sk * 64 > rows:
            continue
        for with_rs in (True, False):
            f = lambda: K.gemm(dy, x, a_kmajor=False, b_kmajor=False, out=dst, accumulate=True, split_k=sk, a_rowsum=rs if with_rs else None)  # noqa: E731
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            torch.cuda.synchronize()
            res.append(f"sk{sk}{'+b' if with_rs else ''} {e0.elapsed_time(e1) / 20 * 1e3:.1f}")
    mb = rows * (n + k) * 2 / 1e6
    print(f"dW[{n}x{k}] over {rows} rows ({mb:.0f} MB, {mb / 4e3 * 1e3:.1f} us at 4 TB/s): " + "  ".join(res), flush=True)
