"""The four weight gradients of an encoder block: four split-K launches vs one grouped launch (dev tooling)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402
from cinema_amd import tape as T  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
for rows, shapes in ((10960, [(2304, 768), (768, 768), (3072, 768), (768, 3072)]), (32848, [(512, 512), (512, 512), (2048, 512), (512, 2048)])):
    probs = []
    for n, k in shapes:
        dy = (torch.randn(rows, n, device=dev) * 0.5).to(torch.bfloat16)
        x = (torch.randn(rows, k, device=dev) * 0.5).to(torch.bfloat16)
        probs.append((dy, x, torch.zeros(n, k, device=dev), torch.zeros(n, device=dev)))

    def single():
        for dy, x, dst, b in probs:
            K.gemm(dy, x, a_kmajor=False, b_kmajor=False, out=dst, accumulate=True, split_k=T._split_k(rows, dy.shape[1], x.shape[1]), a_rowsum=b)

    t1 = timeit(single)
    t2 = timeit(lambda: K.gemm_wgrad_grouped(probs))
    fl = sum(2.0 * rows * n * k for n, k in shapes)
    tiles = sum(((n + 127) // 128) * ((k + 127) // 128) for n, k in shapes)
    print(f"rows {rows}, {tiles} tiles: 4 split-K launches {t1 * 1e6:.1f} us ({fl / t1 / 1e12:.0f} TF) | grouped {t2 * 1e6:.1f} us ({fl / t2 / 1e12:.0f} TF)", flush=True)
