"""Which Python lines launch torch (ATen) kernels / device copies inside one training step (dev tooling): torch.profiler with stacks."""
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
for _ in range(5):
    step(batch, 0.75)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    step(batch, 0.75)
    torch.cuda.synchronize()
cnt: Counter = Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    frame = next((f for f in ev.stack if "/cinema_amd/" in f or "bench.py" in f), ev.stack[0] if ev.stack else "?")
    cnt[(ev.name, frame.split("/root/repo/")[-1] if "/root/repo/" in frame else frame)] += 1
for (name, frame), n in cnt.most_common(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    print(f"{n:5d}  {name:28s} {frame}")
