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
