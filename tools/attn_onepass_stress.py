"""Stress of the key-split one-pass attention backward (csrc/attention.hip attn_bwd_onepass_mfma): N launches on the same operands - several workgroups per (batch, head),
several passes, the last-arriver sum - every result compared BIT for bit with the first (a stale or torn partial sum would show up as a difference), with a streaming
kernel on a second stream beside every third launch; tickets checked at the end (dev tooling).   python tools/attn_onepass_stress.py [launches]"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

os.environ["CINEMA_ATTN_ONEPASS"] = "1"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
bad = 0
for name, (b, heads, tq, tk, g) in {"config 4 shape, 5 workgroups x 3 passes": (4, 12, 3073, 3073, 0), "config 5 shape, 2 x 4 passes": (8, 16, 1537, 1537, 0),
                                    "13 workgroups, one pass": (2, 2, 3073, 3073, 13)}.items():
    hd = 64
    c = heads * hd
    q = (torch.randn(b, tq, c, device="cuda") * 0.5).to(torch.bfloat16)
    kv = (torch.randn(b, tk, 2 * c, device="cuda") * 0.5).to(torch.bfloat16)
    k, v = kv[..., :c], kv[..., c:]
    o, lse = K.attention_fwd(q, k, v, heads, hd**-0.5)
    d_o = (torch.randn(b, tq, c, device="cuda") * 0.5).to(torch.bfloat16)
    if g:
        os.environ["CINEMA_ATTN_ONEPASS_G"] = str(g)
    else:
        os.environ.pop("CINEMA_ATTN_ONEPASS_G", None)
    ref = None
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, device="cuda")
    for it in range(n):
        dq, dkv = torch.full_like(q, float("nan")), torch.full_like(kv, float("nan"))
        if it % 3 == 0:
            with torch.cuda.stream(side):
                junk.mul_(1.0001)
        K.attention_bwd(q, k, v, o, d_o, lse, heads, hd**-0.5, dq, dkv[..., :c], dkv[..., c:])
        if ref is None:
            ref = (dq.clone(), dkv.clone())
            assert bool(torch.isfinite(dq.float()).all()) and bool(torch.isfinite(dkv.float()).all())
        elif not (torch.equal(dq, ref[0]) and torch.equal(dkv, ref[1])):
            bad += 1
    torch.cuda.synchronize()
    print(f"{name}: {n} launches, mismatches so far {bad}", flush=True)
tickets = sum(int(t.abs().sum()) for t in K._ATTN_COUNTERS.values())  # noqa: SLF001
print("ATTN ONE-PASS STRESS", "OK" if bad == 0 and tickets == 0 else "FAILED", {"mismatches": bad, "tickets_left": tickets})
sys.exit(0 if bad == 0 and tickets == 0 else 1)
