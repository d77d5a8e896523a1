"""Timing ablations of the one-pass attention backward (CINEMA_ATTN_ONEPASS_DBG bits; results invalid): python tools/bench_attn_dbg.py"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

b, heads, hd, tq, tk = 16, 12, 64, int(os.environ.get('TQ', '685')), int(os.environ.get('TK', '685'))
c = heads * hd
dev = "cuda"
os.environ["CINEMA_ATTN_ONEPASS"] = "1"
q = (torch.randn(b, tq, c, device=dev) * 0.5).to(torch.bfloat16)
kv = (torch.randn(b, tk, 2 * c, device=dev) * 0.5).to(torch.bfloat16)
k, v = kv[..., :c], kv[..., c:]
scale = hd**-0.5
o, lse = K.attention_fwd(q, k, v, heads, scale)
d_o = (torch.randn(b, tq, c, device=dev) * 0.5).to(torch.bfloat16)
dq, dkv = torch.empty_like(q), torch.empty_like(kv)
for rnd in range(2):
    for dbg in [int(v) for v in os.environ.get('DBG_LIST', '0,32,64,96,128,31,63,127,255').split(',')]:
        os.environ["CINEMA_ATTN_ONEPASS_DBG"] = str(dbg)
        for _ in range(2):
            K.attention_bwd(q, k, v, o, d_o, lse, heads, scale, dq, dkv[..., :c], dkv[..., c:])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            K.attention_bwd(q, k, v, o, d_o, lse, heads, scale, dq, dkv[..., :c], dkv[..., c:])
        e1.record()
        torch.cuda.synchronize()
        print(f"dbg {dbg:2d}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us", flush=True)
