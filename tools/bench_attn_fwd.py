"""Attention forward at the shapes of the BASELINE configs (dev tooling; CINEMA_LIB=path loads a variant build).   python tools/bench_attn_fwd.py [name ...]"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

if os.environ.get("CINEMA_LIB"):
    K._LIB_PATH = Path(os.environ["CINEMA_LIB"]).resolve()  # noqa: SLF001

SHAPES = {  # name: (batch, heads, head_dim, queries, keys)
    "dec2": (16, 16, 32, 2053, 685), "enc2": (16, 12, 64, 685, 685), "enc4": (4, 12, 64, 3073, 3073), "enc5": (8, 16, 64, 1537, 1537), "dec5": (8, 16, 32, 5185, 1729),
}
dev = "cuda"
for name in (sys.argv[1:] or list(SHAPES)):
    b, heads, hd, tq, tk = SHAPES[name]
    c = heads * hd
    q = (torch.randn(b, tq, c, device=dev) * 0.5).to(torch.bfloat16)
    kv = (torch.randn(b, tk, 2 * c, device=dev) * 0.5).to(torch.bfloat16)
    k, v = kv[..., :c], kv[..., c:]
    scale = hd**-0.5
    flops = 4.0 * b * heads * tq * tk * hd
    ts = []
    for rnd in range(3):
        for _ in range(3):
            K.attention_fwd(q, k, v, heads, scale)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            K.attention_fwd(q, k, v, heads, scale)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e-3)
    t = sorted(ts)[1]
    print(f"{name} b{b} h{heads} hd{hd} {tq}x{tk}: {t * 1e6:8.1f} us  {flops / t / 1e12:7.1f} TF", flush=True)
