"""Attention kernels at the step's two shapes (encoder self-attention, decoder cross-attention): forward / backward us and TFLOP/s (dev tooling)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
for name, b, heads, tq, tk, hd in (("encoder self", 16, 12, 685, 685, 64), ("decoder cross", 16, 16, 2053, 684, 32), ("aligned 2048x704", 16, 16, 2048, 704, 32),
                                      ("large encoder", 8, 16, 1537, 1537, 64), ("large decoder", 8, 16, 5185, 1728, 32)):
    c = heads * hd
    q = (torch.randn(b * tq, c, device=dev) * 0.5).to(torch.bfloat16)
    k = (torch.randn(b * tk, c, device=dev) * 0.5).to(torch.bfloat16)
    v = (torch.randn(b * tk, c, device=dev) * 0.5).to(torch.bfloat16)
    do = (torch.randn(b * tq, c, device=dev) * 0.5).to(torch.bfloat16)
    scale = hd ** -0.5
    o, lse = K.attention_fwd(q.view(b, tq, c), k.view(b, tk, c), v.view(b, tk, c), heads, scale)
    dq, dk, dv = torch.empty_like(q).view(b, tq, c), torch.empty_like(k).view(b, tk, c), torch.empty_like(v).view(b, tk, c)
    import os
    os.environ["CINEMA_ATTN_FWD_V2"] = "0"
    tf0 = timeit(lambda: K.attention_fwd(q.view(b, tq, c), k.view(b, tk, c), v.view(b, tk, c), heads, scale))
    os.environ["CINEMA_ATTN_FWD_V2"] = "1"
    tf = timeit(lambda: K.attention_fwd(q.view(b, tq, c), k.view(b, tk, c), v.view(b, tk, c), heads, scale))
    tb = timeit(lambda: K.attention_bwd(q.view(b, tq, c), k.view(b, tk, c), v.view(b, tk, c), o, do.view(b, tq, c), lse, heads, scale, dq, dk, dv))
    fl = 4.0 * b * heads * tq * tk * hd
    print(f"{name:18s} b{b} h{heads} tq{tq} tk{tk} hd{hd}: fwd {tf * 1e6:7.1f} us ({fl / tf / 1e12:5.0f} TF; eager rescale + VALU sums {tf0 * 1e6:7.1f} us) | bwd {tb * 1e6:7.1f} us ({2.5 * fl / tb / 1e12:5.0f} TF)", flush=True)
