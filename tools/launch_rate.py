"""Host cost per kernel launch: a C loop of empty launches, the same through ctypes one by one, and a small real kernel through the Python wrapper."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

lib = K.load()
x = torch.zeros(1024, device="cuda")
torch.cuda.synchronize()
st = K._stream()
for n in (2000, 20000):
    lib.cinema_launch_probe(200, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lib.cinema_launch_probe(n, st)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"C loop, {n} empty launches: {1e6 * (t1 - t0) / n:.2f} us/launch to enqueue, {1e6 * (t2 - t0) / n:.2f} us/launch until the GPU is done")
n = 5000
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    lib.cinema_launch_probe(1, st)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"ctypes, one empty launch per call: {1e6 * (t1 - t0) / n:.2f} us/launch")
t0 = time.perf_counter()
for _ in range(n):
    K.scale(x, 1.0)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"K.scale wrapper (torch.empty + checks + ctypes + launch): {1e6 * (t1 - t0) / n:.2f} us/launch")
t0 = time.perf_counter()
for _ in range(n):
    torch.empty(1024, device="cuda")
t1 = time.perf_counter()
print(f"torch.empty: {1e6 * (t1 - t0) / n:.2f} us")
