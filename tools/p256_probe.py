"""Runs the persistent 256x256 GEMM a few times per operand layout at one shape (for rocprofv3 counter passes: tools/gpu_pmc_p256.sh).
   python tools/p256_probe.py [M N K] [--old]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("-")]
m, n, k = (int(a) for a in args[:3]) if len(args) >= 3 else (4096, 4096, 4096)
old = "--old" in sys.argv
dev = "cuda"
x = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
dy = (torch.randn(m, n, device=dev) * 0.5).to(torch.bfloat16)
y = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
dx = torch.empty(m, k, dtype=torch.bfloat16, device=dev)
dw = torch.zeros(n, k, dtype=torch.float32, device=dev)
kw = {} if old else dict(p256=0, split_k=1)
for _ in range(3):
    K.gemm(x, w, out=y, **kw)
    K.gemm(dy, w, a_kmajor=True, b_kmajor=False, out=dx, **kw)
    if old:
        K.gemm(dy, x, a_kmajor=False, b_kmajor=False, out=dw, accumulate=True, split_k=1)
    else:
        K.gemm_wgrad_grouped([(dy, x, dw, None)], p256=True)
torch.cuda.synchronize()
