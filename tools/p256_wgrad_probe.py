"""The grouped weight gradients of one transformer block on the persistent 256x256 kernel, a few launches (for counter passes: tools/gpu_pmc_traffic_cmd.sh).
   python tools/p256_wgrad_probe.py enc|dec [whole]      (whole: split_k = 1, whole-K tiles, no in-launch reduction)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

BLOCKS = {
    "enc": [(10960, 2304, 768), (10960, 768, 768), (10960, 3072, 768), (10960, 768, 3072)],
    "cal": [(262144, 256, 256)],   # one output tile: every operand byte is needed exactly once, no reuse possible (calibrates the counter arithmetic)
    "dec": [(32848, 512, 512), (10944, 1024, 512), (32848, 512, 512), (32848, 2048, 512), (32848, 512, 2048)],
}
BLOCKS["enc2"] = BLOCKS["enc"] * 2   # the step's launch: two ViT-Base encoder blocks = 216 whole-K tiles, one per workgroup
BLOCKS["dec2"] = [(32848, 512, 512), (32848, 512, 512), (32848, 2048, 512), (32848, 512, 2048)] * 2
which = sys.argv[1] if len(sys.argv) > 1 else "enc"
whole = "whole" in sys.argv
probs = []
for rows, n, k in BLOCKS[which]:
    dy = (torch.randn(rows, n, device="cuda") * 0.5).to(torch.bfloat16)
    x = (torch.randn(rows, k, device="cuda") * 0.5).to(torch.bfloat16)
    probs.append((dy, x, torch.zeros(n, k, dtype=torch.float32, device="cuda"), torch.zeros(n, dtype=torch.float32, device="cuda")))
alg = sum(2 * r * (n + k) + 8 * n * k + 8 * n for r, n, k in BLOCKS[which])
print(which, "whole-K" if whole else "split", "algorithmic bytes per launch", alg)
for _ in range(4):
    K.gemm_wgrad_grouped(probs, p256=True, split_k=1 if whole else 0)
torch.cuda.synchronize()
