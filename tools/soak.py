"""Soak test of the recorded MAE step: N replayed steps on two resident batches; the loss stays finite and falls, the persistent GEMM's tile counters and
error word (and the optional in-launch tail counters) are zero at the end, peak memory does not grow (dev tooling).   python tools/soak.py [steps]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from cinema_amd import CineMA  # noqa: E402
from cinema_amd import hip as K  # noqa: E402
from cinema_amd.optim import TrainStep  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
kw = bench.base_kwargs("base")
torch.manual_seed(0)
model = CineMA(**kw).to("cuda")
step = TrainStep(model, lr=2e-4, replay=True)
batches = [bench.synthetic_batch(kw, 16, s, "cuda") for s in (1, 2)]
losses, mem = [], []
t0 = time.time()
for i in range(n):
    loss, gn, _ = step(batches[i & 1], 0.75)
    if i % 100 == 0 or i == n - 1:
        losses.append(float(loss))
        mem.append(torch.cuda.max_memory_reserved() / 2**30)
        print(f"step {i:5d} loss {losses[-1]:.5f} grad_norm {float(gn):.4f} reserved {mem[-1]:.2f} GiB  {time.time() - t0:6.1f} s", flush=True)
torch.cuda.synchronize()
bad = []
for key, ws in K._P256_WS.items():
    head = ws[:16384].view(torch.int32)
    if int(head.abs().sum()) != 0:
        bad.append(("p256 counters / error word", key, int(head.abs().sum())))
for key, t in K._TAIL_COUNTERS.items():
    if int(t.abs().sum()) != 0:
        bad.append(("tail counters", key))
ok = all(map(lambda v: v == v and abs(v) < 1e4, losses)) and losses[-1] < losses[0] and mem[-1] <= mem[1] + 0.01 and not bad
print("SOAK", "OK" if ok else "FAILED", {"first": losses[0], "last": losses[-1], "mem": (mem[1], mem[-1]), "bad": bad})
sys.exit(0 if ok else 1)
