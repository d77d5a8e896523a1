"""Which module lines issue a given library call during one training step (dev tooling): python tools/who_calls.py cinema_cast [cinema_row_copy ...]"""
import sys
import traceback
from collections import Counter
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from cinema_amd import CineMA  # noqa: E402
from cinema_amd import hip as K  # noqa: E402
from cinema_amd.optim import TrainStep  # noqa: E402

names = sys.argv[1:] or ["cinema_cast"]
kw = bench.base_kwargs("base")
torch.manual_seed(0)
model = CineMA(**kw).to("cuda")
step = TrainStep(model)
batch = bench.synthetic_batch(kw, 16, 1, "cuda")
for _ in range(2):
    step(batch, 0.75)
lib = K.load()
counts = {n: Counter() for n in names}
for n in names:
    entry = getattr(lib, n)
    orig = entry.fn

    def wrapped(*a, _orig=orig, _n=n):
        fr = [f for f in traceback.extract_stack() if "/cinema_amd/" in f.filename and "/cinema_amd/hip/" not in f.filename]
        key = " <- ".join(f"{f.filename.split('/cinema_amd/')[-1]}:{f.lineno}" for f in fr[-3:][::-1])
        counts[_n][key] += 1
        return _orig(*a)

    entry.fn = wrapped
T = __import__("cinema_amd.tape", fromlist=["x"])
step(batch, 0.75)
torch.cuda.synchronize()
for n in names:
    print(f"== {n}: {sum(counts[n].values())} calls")
    for k, v in counts[n].most_common(14):
        print(f"   {v:4d}  {k}")
