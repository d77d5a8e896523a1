"""Do small kernels on several HIP streams run concurrently? (dev tooling)  N launches of a one-workgroup ~10 us kernel (cinema_mfma_probe) on
one stream vs round-robin over 2 / 4 / 8 streams: if the GPU-done time per launch drops with the stream count, they overlap."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

lib = K.load()
out = torch.zeros(4, device="cuda")
streams = [torch.cuda.Stream().cuda_stream for _ in range(8)]
N = 4000
for iters in (40, 200):
    for n_streams in (1, 2, 4, 8, 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            lib.cinema_mfma_probe(1, iters, out.data_ptr(), streams[i % n_streams])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"kernel of {iters} iterations, {n_streams} stream(s): enqueue {1e6 * (t1 - t0) / N:.2f} us/launch, GPU done after {1e6 * (t2 - t0) / N:.2f} us/launch", flush=True)
