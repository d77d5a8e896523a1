"""Step time with the main chain on a HIGH-priority HIP stream (the weight-gradient / long-axis streams stay at normal priority) vs everything at normal priority (dev tooling).
   python tools/prio_ab.py hi|def     one mode per process: the recorded launch list binds the stream handles"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from cinema_amd import CineMA  # noqa: E402
from cinema_amd.optim import TrainStep  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "def"
kw = bench.base_kwargs("base")
torch.manual_seed(0)
model = CineMA(**kw).to("cuda")
batches = [bench.synthetic_batch(kw, 16, i, "cuda") for i in range(2)]
torch.cuda.synchronize()
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
st = torch.cuda.Stream(priority=-1) if mode == "hi" else torch.cuda.current_stream()
with torch.cuda.stream(st):
    step = TrainStep(model, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05, clip_grad=5.0, replay=True)
    for i in range(25):
        step(batches[i % 2], 0.75)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(30):
        loss, gnorm, _ = step(batches[i % 2], 0.75)
    torch.cuda.synchronize()
    print(f"main chain on a {'HIGH-priority' if mode == 'hi' else 'normal'} stream (priority range {lo}..{hi}, stream priority {st.priority}): "
          f"{1e3 * (time.perf_counter() - t0) / 30:.2f} ms/step, loss {float(loss):.5f}", flush=True)
