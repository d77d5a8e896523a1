"""A/B of the attention backward forms at the decoder shape of the MAE step (dev tooling): python tools/bench_attn_bwd.py"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

b, heads, hd, tq, tk = 16, 16, 32, 2053, 684
c = heads * hd
dev = "cuda"
q = (torch.randn(b, tq, c, device=dev) * 0.5).to(torch.bfloat16)
kv = (torch.randn(b, tk, 2 * c, device=dev) * 0.5).to(torch.bfloat16)
k, v = kv[..., :c], kv[..., c:]
scale = hd**-0.5
o, lse = K.attention_fwd(q, k, v, heads, scale)
d_o = (torch.randn(b, tq, c, device=dev) * 0.5).to(torch.bfloat16)
dq, dkv = torch.empty_like(q), torch.empty_like(kv)
flops = 2.5 * 4.0 * b * heads * tq * tk * hd
for rnd in range(3):
    for form in ("0", "1"):
        os.environ["CINEMA_ATTN_FUSED"] = form
        for _ in range(2):
            K.attention_bwd(q, k, v, o, d_o, lse, heads, scale, dq, dkv[..., :c], dkv[..., c:])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            K.attention_bwd(q, k, v, o, d_o, lse, heads, scale, dq, dkv[..., :c], dkv[..., c:])
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        print(f"form {'one-pass' if form == '1' else 'two-kernel'}: {t * 1e6:8.1f} us  {flops / t / 1e12:7.1f} TF-equivalent (5 matmuls of the two-kernel count)", flush=True)
