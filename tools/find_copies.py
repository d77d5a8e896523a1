"""Which host lines launch aten copy/fill/elementwise kernels inside one training step (dev tooling)."""
import sys
from collections import Counter
from pathlib import Path

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from cinema_amd import CineMA  # noqa: E402
from cinema_amd.optim import TrainStep  # noqa: E402

kw = bench.base_kwargs("base")
torch.manual_seed(0)
model = CineMA(**kw).to("cuda")
step = TrainStep(model)
batch = bench.synthetic_batch(kw, 16, 1, "cuda")
for _ in range(2):
    step(batch, 0.75)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step(batch, 0.75)
    torch.cuda.synchronize()
cnt = Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::mul", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy",
                   "aten::index", "aten::sort", "aten::argsort", "aten::arange", "aten::zeros", "aten::cat"):
        frames = [f for f in (ev.stack or []) if "cinema_amd" in f or "bench.py" in f]
        where = frames[0].strip() if frames else "?"
        shapes = str(ev.input_shapes)[:60]
        cnt[(ev.name, where[-70:], shapes)] += 1
for (name, where, shapes), n in cnt.most_common(70):
    print(f"{n:4d} {name:18s} {where:72s} {shapes}")
