"""LayerNorm forward / backward timings at the step's shapes (dev tooling)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402
from tools.bench_p256 import bench  # noqa: E402

dev = "cuda"
for rows, c in ((10960, 768), (32848, 512), (13824, 1024), (41480, 512)):
    x = torch.randn(rows, c, device=dev)
    g, b = torch.randn(c, device=dev), torch.randn(c, device=dev)
    dy = (torch.randn(rows, c, device=dev) * 0.1).to(torch.bfloat16)
    res = torch.randn(rows, c, device=dev)
    y16, _, mean, rstd = K.layernorm_fwd(x, g, b, 1e-6)
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    r = bench({"fwd": lambda: K.layernorm_fwd(x, g, b, 1e-6),
               "bwd": lambda: K.layernorm_bwd(dy, x, g, b, mean, rstd, dx_residual=res, want_f32=True, want_bf16=True, dgamma=dg, dbeta=db, deferred=[])}, iters=10)
    fb, bb = rows * c * (4 + 2), rows * c * (4 + 2 + 4 + 4 + 2)
    print(f"rows {rows} c {c}: fwd {r['fwd'] * 1e6:6.1f} us ({fb / r['fwd'] / 1e12:4.2f} TB/s) | bwd {r['bwd'] * 1e6:6.1f} us ({bb / r['bwd'] / 1e12:4.2f} TB/s)", flush=True)

# narrow rows (conv stems, ConvUNetR decoder): (rows, c, x bf16?, dy bf16?, residual?, act)
print("narrow rows: rows c | x dy res act | bwd us (TB/s)")
for rows, c, xb, dyb, has_res, act in ((147456, 64, False, True, True, 0), (36864, 128, False, True, True, 0), (196608, 64, False, True, True, 0), (49152, 128, False, True, True, 0),
                                       (3145728, 32, True, False, False, 1), (786432, 64, True, False, False, 1), (196608, 128, True, False, False, 1), (786432, 64, False, True, True, 0)):
    x = torch.randn(rows, c, device=dev)
    x = x.to(torch.bfloat16) if xb else x
    g, b = torch.randn(c, device=dev), torch.randn(c, device=dev)
    dy = torch.randn(rows, c, device=dev) * 0.1
    dy = dy.to(torch.bfloat16) if dyb else dy
    res = torch.randn(rows, c, device=dev) if has_res else None
    _, _, mean, rstd = K.layernorm_fwd(x, g, b, 1e-6, act=act)
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    want16 = not xb
    r = bench({"bwd": lambda: K.layernorm_bwd(dy, x, g, b, mean, rstd, act=act, dx_residual=res, want_f32=not xb, want_bf16=True, dgamma=dg, dbeta=db, deferred=[])}, iters=10)
    bb = rows * c * ((2 if xb else 4) + (2 if dyb else 4) + (4 if has_res else 0) + (0 if xb else 4) + 2)
    print(f"{rows:8d} {c:4d} | {'bf16' if xb else 'f32 '} {'bf16' if dyb else 'f32 '} {int(has_res)} {act} | {r['bwd'] * 1e6:7.1f} us ({bb / r['bwd'] / 1e12:4.2f} TB/s)", flush=True)

# the e4m3-path form of the backward (8-bit copy of dx under a delayed per-tensor scale, no bf16 copy, column sums of dx as a third partial row) beside the bf16-path form
from cinema_amd import tape as T  # noqa: E402

print("fp8-path LayerNorm backward: rows c | bf16-path us | q8 us | q8 + dcol us")
sites = T.Fp8Sites(torch.device(dev, torch.cuda.current_device()))
for rows, c in ((13832, 1024), (41480, 512), (10960, 768)):
    x = torch.randn(rows, c, device=dev)
    g, b = torch.randn(c, device=dev), torch.randn(c, device=dev)
    dy = (torch.randn(rows, c, device=dev) * 0.1).to(torch.bfloat16)
    res = torch.randn(rows, c, device=dev)
    _, _, mean, rstd = K.layernorm_fwd(x, g, b, 1e-6)
    dg, db, dc = torch.zeros(c, device=dev), torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    s = sites.site(("bench", rows, c))
    K.layernorm_bwd(dy, x, g, b, mean, rstd, dx_residual=res, want_f32=True, want_bf16=True, dgamma=dg, dbeta=db, deferred=[], q8=s)
    sites.update()
    r = bench({"bf16": lambda: K.layernorm_bwd(dy, x, g, b, mean, rstd, dx_residual=res, want_f32=True, want_bf16=True, dgamma=dg, dbeta=db, deferred=[]),
               "q8": lambda: K.layernorm_bwd(dy, x, g, b, mean, rstd, dx_residual=res, want_f32=True, want_bf16=False, dgamma=dg, dbeta=db, deferred=[], q8=s),
               "q8dcol": lambda: K.layernorm_bwd(dy, x, g, b, mean, rstd, dx_residual=res, want_f32=True, want_bf16=False, dgamma=dg, dbeta=db, deferred=[], q8=s, q8_colsum=dc)}, iters=10)
    print(f"{rows:6d} {c:5d} | {r['bf16'] * 1e6:6.1f} | {r['q8'] * 1e6:6.1f} | {r['q8dcol'] * 1e6:6.1f}", flush=True)
