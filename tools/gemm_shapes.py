"""Per-shape GEMM table of one training step (single stream, HIP events around every launch): where the GEMM time goes (dev tooling)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from cinema_amd import CineMA  # noqa: E402
from cinema_amd import hip as K  # noqa: E402
from cinema_amd import tape as T  # noqa: E402
from cinema_amd.optim import TrainStep  # noqa: E402

kw = bench.base_kwargs("base")
torch.manual_seed(0)
model = CineMA(**kw).to("cuda")
step = TrainStep(model)
batch = bench.synthetic_batch(kw, 16, 1, "cuda")
T.SIDE_WGRAD = False
for _ in range(12):
    step(batch, 0.75)
N = 4
K.GEMM_PROFILE = []
for _ in range(N):
    step(batch, 0.75)
torch.cuda.synchronize()
prof, K.GEMM_PROFILE = K.GEMM_PROFILE, None
agg: dict = {}
for kind, flops, e0, e1, shape, *_ in prof:
    a = agg.setdefault((kind, shape), [0.0, 0.0, 0])
    a[0] += flops
    a[1] += e0.elapsed_time(e1) * 1e-3
    a[2] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values()) / N * 1e3
print(f"all GEMM launches: {tot:.2f} ms/step")
print(f"{'kernel':38s} {'m':>6s} {'n':>5s} {'k':>6s} aK bK sk  n/step  us/launch    TF   ms/step")
for (kind, (m, n, k, ak, bk, sk, *_)), (fl, secs, cnt) in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 60]:
    print(f"{K.GEMM_KERNEL_NAMES[kind]:38s} {m:6d} {n:5d} {k:6d} {ak:2d} {bk:2d} {sk:2d} {cnt // N:7d} {secs / cnt * 1e6:10.1f} {fl / secs / 1e12:6.0f} {secs / N * 1e3:8.3f}")
