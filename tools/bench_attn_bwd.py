"""A/B of the attention backward forms (one-pass vs the dQ + dK/dV kernel pair) at the shapes of the BASELINE configs (dev tooling):
   python tools/bench_attn_bwd.py [name ...]     names: dec2 enc2 enc4 enc5 dec5"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

SHAPES = {  # name: (batch, heads, head_dim, queries, keys)
    "dec2": (16, 16, 32, 2053, 685),   # config 2, decoder cross attention
    "enc2": (16, 12, 64, 685, 685),    # config 2, encoder
    "enc4": (4, 12, 64, 3073, 3073),   # config 4, encoder at every token
    "enc5": (8, 16, 64, 1537, 1537),   # config 5, ViT-Large encoder
    "dec5": (8, 16, 32, 5185, 1729),   # config 5, decoder cross attention
}
dev = "cuda"
for name in (sys.argv[1:] or list(SHAPES)):
    b, heads, hd, tq, tk = SHAPES[name]
    c = heads * hd
    q = (torch.randn(b, tq, c, device=dev) * 0.5).to(torch.bfloat16)
    kv = (torch.randn(b, tk, 2 * c, device=dev) * 0.5).to(torch.bfloat16)
    k, v = kv[..., :c], kv[..., c:]
    scale = hd**-0.5
    o, lse = K.attention_fwd(q, k, v, heads, scale)
    d_o = (torch.randn(b, tq, c, device=dev) * 0.5).to(torch.bfloat16)
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    flops = 2.5 * 4.0 * b * heads * tq * tk * hd
    os.environ["CINEMA_ATTN_ONEPASS"] = "1"
    forms = [("two-kernel", {"CINEMA_ATTN_FUSED": "0"}), ("one-pass", {"CINEMA_ATTN_FUSED": "1"})]
    if hd == 64:
        forms += [(f"one-pass G={g}", {"CINEMA_ATTN_FUSED": "1", "CINEMA_ATTN_ONEPASS_G": str(g)}) for g in os.environ.get("BENCH_G", "").split(",") if g]
    for rnd in range(2):
        for label, env in forms:
            os.environ.pop("CINEMA_ATTN_ONEPASS_G", None)
            os.environ.update(env)
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
            print(f"{name} b{b} h{heads} hd{hd} {tq}x{tk}  {label:16s}: {t * 1e6:8.1f} us  {flops / t / 1e12:7.1f} TF-equivalent (5 matmuls)", flush=True)
