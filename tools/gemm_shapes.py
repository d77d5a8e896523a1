"""Per-shape GEMM time inside one real training step (dev tooling): python tools/gemm_shapes.py"""
import sys
from collections import defaultdict
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from cinema_amd import CineMA  # noqa: E402
from cinema_amd import hip as K  # noqa: E402
from cinema_amd.optim import TrainStep  # noqa: E402

kw = bench.base_kwargs("base")
torch.manual_seed(0)
model = CineMA(**kw).to("cuda")
step = TrainStep(model)
batch = bench.synthetic_batch(kw, 16, 1, "cuda")
for _ in range(2):
    step(batch, 0.75)
K.GEMM_PROFILE = []
step(batch, 0.75)
torch.cuda.synchronize()
prof, K.GEMM_PROFILE = K.GEMM_PROFILE, None
agg = defaultdict(lambda: [0.0, 0, 0.0])
for kind, flops, e0, e1, shape in prof:
    a = agg[(kind,) + shape]
    a[0] += e0.elapsed_time(e1) * 1e-3
    a[1] += 1
    a[2] += flops
tot = sum(v[0] for v in agg.values())
print(f"total GEMM time {tot*1e3:.2f} ms in {sum(v[1] for v in agg.values())} launches")
print("kind  M        N     K     aK bK split | calls  total_ms  avg_us   TF")
for key, (t, n, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    kind, m, nn, k, ak, bk, sp = key
    print(f"{kind:4d} {m:8d} {nn:5d} {k:6d}  {ak}  {bk}  {sp:4d} | {n:5d} {t*1e3:8.3f} {t/n*1e6:8.1f} {fl/t/1e12:7.1f}")
