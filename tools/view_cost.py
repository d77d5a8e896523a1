"""Cost of the long-axis views: the recorded step with SAX only / SAX + 1 / + 2 / + 3 long-axis views (dev tooling)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from cinema_amd import CineMA  # noqa: E402
from cinema_amd.optim import TrainStep  # noqa: E402

kw = bench.base_kwargs("base")
full = bench.synthetic_batch(kw, 16, 1, "cuda")
torch.manual_seed(0)
model = CineMA(**kw).to("cuda")
step = TrainStep(model, replay=True)
names = list(full)
for n in (1, 2, 3, 4, 1, 4):
    batch = {k: full[k] for k in names[:n]}
    for _ in range(15):
        step(batch, 0.75)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        step(batch, 0.75)
    torch.cuda.synchronize()
    rec = step._recorded[(0.75, tuple((k, tuple(v.shape), v.dtype) for k, v in batch.items()))]
    print(f"{n} view(s) {names[:n]}: {1e3 * (time.perf_counter() - t0) / 30:.2f} ms/step, {rec.n_launches} launches", flush=True)
