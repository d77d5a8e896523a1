"""LayerNorm forward at the steps' shapes, narrow conv rows included (dev tooling; CINEMA_LIB=path loads a variant build).
   python tools/bench_ln_fwd.py"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

if os.environ.get("CINEMA_LIB"):
    K._LIB_PATH = Path(os.environ["CINEMA_LIB"]).resolve()  # noqa: SLF001
from tools.bench_p256 import bench  # noqa: E402

dev = "cuda"
# (rows, c, x bf16?, act, bf16 out?)  config-2 stems (visible voxels), config-4 decoder levels, transformer widths
SHAPES = ((589824, 32, True, 1, True), (147456, 64, True, 1, True), (147456, 64, False, 0, True), (36864, 128, False, 0, True),
          (3145728, 32, True, 1, True), (786432, 64, True, 1, True), (196608, 128, True, 1, True), (49152, 256, True, 1, True),
          (32848, 512, False, 0, True), (10960, 768, False, 0, True), (13824, 1024, False, 0, True))
for rows, c, xb, act, _ in SHAPES:
    x = torch.randn(rows, c, device=dev)
    x = x.to(torch.bfloat16) if xb else x
    g, b = torch.randn(c, device=dev), torch.randn(c, device=dev)
    r = bench({"fwd": lambda: K.layernorm_fwd(x, g, b, 1e-6, act=act)}, iters=20)
    nbytes = rows * c * ((2 if xb else 4) + 2) + rows * 8
    print(f"{rows:8d} x {c:4d} {'bf16' if xb else 'f32 '} act {act}: {r['fwd'] * 1e6:7.1f} us  {nbytes / r['fwd'] / 1e12:5.2f} TB/s", flush=True)
