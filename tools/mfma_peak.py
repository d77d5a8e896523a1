"""Sustained bf16 MFMA rate of this device with no memory traffic (clock and power limits included): the practical ceiling for the GEMMs."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

lib = K.load()
out = torch.zeros(4, device="cuda")
st = K._stream()
for grid, iters in ((512, 2000), (512, 20000), (1024, 20000), (256, 20000)):
    lib.cinema_mfma_probe(grid, 100, out.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.cinema_mfma_probe(grid, iters, out.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    fl = grid * 4 * iters * 16 * 32768.0
    print(f"grid {grid} x 4 waves x {iters} x 16 MFMA 32x32x16 bf16: {ms:.3f} ms -> {fl / ms / 1e9:.0f} TFLOP/s")
