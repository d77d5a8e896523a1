"""How much of the persistent 256x256 weight-gradient launch's time is HBM traffic?  The step's launch (two ViT-Base encoder blocks, 216 whole-K tiles) with its
real operands against the same launch with every operand ROW aliased to one row (leading dimension 0: the operands are 4.6 KB, L2-resident, no HBM stream at all;
the arithmetic, the LDS-DMA issue and the epilogue are unchanged).   python tools/p256_traffic_sensitivity.py [enc2|dec2]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

BLOCKS = {
    "enc2": [(10960, 2304, 768), (10960, 768, 768), (10960, 3072, 768), (10960, 768, 3072)] * 2,
    "dec2": [(32848, 512, 512), (32848, 512, 512), (32848, 2048, 512), (32848, 512, 2048)] * 2,
}
which = sys.argv[1] if len(sys.argv) > 1 else "enc2"


def build(alias: bool) -> list:
    probs = []
    for rows, n, k in BLOCKS[which]:
        if alias:
            dy = (torch.randn(1, n, device="cuda") * 0.5).to(torch.bfloat16).expand(rows, n)
            x = (torch.randn(1, k, device="cuda") * 0.5).to(torch.bfloat16).expand(rows, k)
        else:
            dy = (torch.randn(rows, n, device="cuda") * 0.5).to(torch.bfloat16)
            x = (torch.randn(rows, k, device="cuda") * 0.5).to(torch.bfloat16)
        probs.append((dy, x, torch.zeros(n, k, dtype=torch.float32, device="cuda"), None))
    return probs


flop = sum(2.0 * r * n * k for r, n, k in BLOCKS[which])
for rnd in range(3):
    for alias in (False, True):
        probs = build(alias)
        for _ in range(3):
            K.gemm_wgrad_grouped(probs, p256=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            K.gemm_wgrad_grouped(probs, p256=True)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100.0
        print(f"{which} round {rnd} {'rows aliased (no HBM stream)' if alias else 'real operands':30s} {us:8.1f} us  {flop / us * 1e-6:7.1f} TF", flush=True)
        del probs
