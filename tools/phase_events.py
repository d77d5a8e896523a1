"""True (unprofiled) main-stream timeline of a replayed MAE step: HIP events recorded on the main stream every N entries of the recorded launch list, the elapsed time
between consecutive events averaged over several steps, and the library calls of each segment (rocprofv3 inflates the small launches; this does not).  Dev tooling.
   python tools/phase_events.py [entries-per-segment]"""
import sys
from collections import Counter
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from cinema_amd import CineMA  # noqa: E402
from cinema_amd.optim import TrainStep  # noqa: E402

seg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
kw = bench.base_kwargs("base")
torch.manual_seed(0)
model = CineMA(**kw).to("cuda")
step = TrainStep(model, lr=1e-3, replay=True)
batches = [bench.synthetic_batch(kw, 16, s, "cuda") for s in (1, 2)]
for i in range(12):
    step(batches[i & 1], 0.75)
rec = next(iter(step._recorded.values()))  # noqa: SLF001
calls = rec.calls
events, new_calls, names = [], [], []
cur: Counter = Counter()
for i, (fn, args) in enumerate(calls):
    if i % seg == 0:
        ev = torch.cuda.Event(enable_timing=True)
        events.append(ev)
        new_calls.append((None, ev.record))
        if i:
            names.append(cur)
            cur = Counter()
    new_calls.append((fn, args))
    cur[getattr(fn, "__name__", "host") if fn is not None else "host"] += 1
end = torch.cuda.Event(enable_timing=True)
events.append(end)
new_calls.append((None, end.record))
names.append(cur)
rec.calls = new_calls
n_steps = 10
acc = [0.0] * (len(events) - 1)
tot, opt = 0.0, 0.0
for i in range(n_steps + 2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    step(batches[i & 1], 0.75)
    e1.record()
    torch.cuda.synchronize()
    if i >= 2:
        for j in range(len(acc)):
            acc[j] += events[j].elapsed_time(events[j + 1])
        tot += e0.elapsed_time(e1)
        opt += end.elapsed_time(e1)
print(f"step {tot / n_steps:.3f} ms (main stream, incl. input copies / mask draw before the list: {(tot - sum(acc) - opt) / n_steps:.3f} ms, list {sum(acc) / n_steps:.3f} ms, clip + AdamW + zero_grad {opt / n_steps:.3f} ms)")
t = 0.0
for j, a in enumerate(acc):
    a /= n_steps
    top = ", ".join(f"{k.replace('cinema_', '')} x{v}" for k, v in names[j].most_common(4))
    print(f"{t:7.3f} +{a:6.3f} ms  entries {j * seg:4d}-{min(len(calls), (j + 1) * seg) - 1:4d}  {top}")
    t += a
