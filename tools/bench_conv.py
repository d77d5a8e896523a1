"""Implicit-GEMM 3x3x3 convolution at the ConvUNetR decoder levels (BASELINE config 4, batch 4): forward / data-gradient and weight-gradient time,
useful TFLOP/s (dev tooling)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402
from cinema_amd.tape import _split_k_conv as _split_k  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = torch.device("cuda")
b = 4
for sp, c in (((256, 256, 12), 32), ((128, 128, 12), 64), ((64, 64, 12), 128), ((32, 32, 12), 256), ((16, 16, 12), 512)):
    x = (torch.randn(b, *sp, c, device=dev) * 0.5).to(torch.bfloat16)
    ld = 27 * c
    w = (torch.randn(c, ld, device=dev) * 0.05).to(torch.bfloat16)
    taps = K.conv_tap_table(c, (3, 3, 3), sp, ld, False, dev)
    rows = b * sp[0] * sp[1] * sp[2]
    fl = 2.0 * rows * ld * c
    tf = timeit(lambda: K.conv_gemm(x, w, taps))
    dy = (torch.randn(rows, c, device=dev) * 0.5).to(torch.bfloat16)
    coords = K.conv_coord_table(b, sp, dev)
    dw = torch.zeros(c, ld, device=dev)
    split = _split_k(rows, c, ld)
    tw = timeit(lambda: K.conv_wgrad(dy, x, taps, coords, dw, split))
    print(f"{sp} c{c:4d}: rows {rows:8d} K {ld:6d} | fwd/dgrad {tf * 1e6:8.1f} us ({fl / tf / 1e12:6.1f} TF) | wgrad {tw * 1e6:8.1f} us ({fl / tw / 1e12:6.1f} TF) [split {split}]", flush=True)
    from cinema_amd.tape import conv_zblock

    zb = conv_zblock(c, (3, 3, 3), sp)
    if zb > 1:  # the z-blocked form the step uses for this level
        wz, _ = K.conv_weight_zblock(w, c, zb, False)
        taps_z = K.conv_tap_table(c, (3, 3, 3), sp, wz.shape[1], False, dev, zb=zb)
        tfz = timeit(lambda: K.conv_gemm(x, wz, taps_z, zb=zb))
        coords_z = K.conv_coord_table(b, sp, dev, zb=zb)
        r = torch.empty(zb * c, wz.shape[1], device=dev)
        sz = _split_k(rows // zb, zb * c, wz.shape[1])
        twz = timeit(lambda: K.conv_wgrad(dy.view(-1, zb * c), x, taps_z, coords_z, r, sz, zb=zb, accumulate=False))
        print(f"    z-blocked x{zb}: rows {rows // zb:8d} K {wz.shape[1]:6d} | fwd/dgrad {tfz * 1e6:8.1f} us ({fl / tfz / 1e12:6.1f} TF useful) | wgrad {twz * 1e6:8.1f} us ({fl / twz / 1e12:6.1f} TF useful) [split {sz}]", flush=True)
