"""Stand-alone timings of the fused conv-stem kernels (csrc/stem.hip, csrc/stem_dw.hip) at BASELINE config 2's short-axis shapes (batch 16, 25 % of the
12 x 12 x 16 tokens kept): stage 1 = 147456 rows x 64 channels (4 x 4 x 1 voxels per token), stage 2 = 36864 rows x 128 channels (2 x 2 x 1).  HIP events around
REPS launches on the current stream; the algorithmic HBM bytes of each kernel and the rate they imply.  Dev tool (GPU box): python tools/bench_stem.py [filter]"""
from __future__ import annotations

import math
import sys

import torch

sys.path.insert(0, ".")
from cinema_amd import hip as K  # noqa: E402
from cinema_amd.convvit import hierarchical_positions  # noqa: E402

DEV = "cuda"
REPS = 20


def timeit(fn) -> float:  # noqa: ANN001
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / REPS * 1e3  # us


def geom_for(batch: int, grid: tuple, block: tuple, ps: list):  # noqa: ANN201
    n_all = math.prod(grid)
    n_keep = n_all // 4
    g = torch.Generator().manual_seed(0)
    keep_pos = torch.stack([torch.randperm(n_all, generator=g)[:n_keep].sort().values for _ in range(batch)])
    keep = (torch.arange(batch)[:, None] * n_all + keep_pos).reshape(-1).to(torch.int32).to(DEV)
    rank = torch.full((batch * n_all,), -1, dtype=torch.int32, device=DEV)
    rank[keep.long()] = torch.arange(keep.numel(), dtype=torch.int32, device=DEV)
    pos = torch.tensor(hierarchical_positions(ps, 1), dtype=torch.int32, device=DEV)
    return K.sparse_geom(batch, grid, block, keep, rank, pos), keep.numel() * math.prod(block)


def main() -> None:
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    torch.manual_seed(0)
    out = []
    for name, c, block, ps in (("stage1", 64, (4, 4, 1), [(1, 1, 1), (2, 2, 1), (2, 2, 1)]), ("stage2", 128, (2, 2, 1), [(1, 1, 1), (2, 2, 1)])):
        geom, rows = geom_for(int(__import__('os').environ.get('BENCH_BATCH', '16')), (12, 12, 16), block, ps)
        h = 4 * c
        f32 = lambda *s: torch.randn(*s, device=DEV)  # noqa: E731
        b16 = lambda *s: torch.randn(*s, device=DEV).to(torch.bfloat16)  # noqa: E731
        x, g2 = f32(rows, c), f32(rows, c)
        d, dh = b16(rows, c), b16(rows, c)
        gam, bet, b1, b2, bf2, bf1 = 1 + 0.1 * f32(c), 0.1 * f32(c), 0.1 * f32(c), 0.1 * f32(c), 0.1 * f32(c), 0.1 * f32(h)
        w1, w2, wf1, wf2 = b16(c, c) * c ** -0.5, b16(c, c) * c ** -0.5, b16(h, c) * c ** -0.5, b16(c, h) * h ** -0.5
        wdw, bdw = 0.2 * f32(c, 1, 5, 5, 5), 0.1 * f32(c)
        x1, _ = K.stem_mlp_fwd(d, x, w2, b2, gam, bet, 1e-6, wf1, bf1, wf2, bf2)
        o = K.stem_mlp_bwd(g2, x1, w2, gam, bet, 1e-6, wf1, bf1, wf2)
        xn, _ = K.stem_ln_linear(x, gam, bet, 1e-6, w1, b1)
        dws = [torch.zeros(c, h, device=DEV), torch.zeros(h, c, device=DEV), torch.zeros(c, c, device=DEV), torch.zeros(c, c, device=DEV)]
        dbs = [torch.zeros(c, device=DEV), torch.zeros(h, device=DEV), torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)]
        probs = [(o["g2_16"], o["a"], dws[0], dbs[0]), (o["dz"], o["xn2"], dws[1], dbs[1]), (o["dx1_16"], d, dws[2], dbs[2]), (dh, xn, dws[3], dbs[3])]
        dwg, dbg = torch.zeros_like(wdw), torch.zeros(c, device=DEV)
        mb = rows * c / 1e6  # one byte per element of a [rows, c] tensor
        cases = [
            ("ln_linear", lambda: K.stem_ln_linear(x, gam, bet, 1e-6, w1, b1), mb * (4 + 2 + 2)),
            ("dw_fwd", lambda: K.sparse_dwconv(d, wdw, bdw, geom), mb * (2 + 2)),
            ("mlp_fwd", lambda: K.stem_mlp_fwd(d, x, w2, b2, gam, bet, 1e-6, wf1, bf1, wf2, bf2), mb * (2 + 4 + 4 + 4)),
            ("mlp_bwd", lambda: K.stem_mlp_bwd(g2, x1, w2, gam, bet, 1e-6, wf1, bf1, wf2), mb * (4 + 4 + 4 + 2 + 2 + 2 + 2 + 8 + 8)),
            ("dw_dgrad", lambda: K.sparse_dwconv(d, wdw, None, geom, flip=True), mb * (2 + 2)),
            ("dw_wgrad", lambda: K.sparse_dwconv_bwd_weight(d, dh, tuple(wdw.shape), dwg, dbg, geom), mb * (2 + 2)),
            ("ln_linear_bwd", lambda: K.stem_ln_linear_bwd(dh, x, g2, gam, 1e-6, w1), mb * (2 + 4 + 4 + 4)),
            ("wgrad", lambda: K.stem_wgrad(probs), mb * 2 * (1 + 4 + 4 + 1 + 1 + 1 + 1 + 1)),
            ("wgrad_p256", lambda: K.gemm_wgrad_grouped(probs, p256=True), mb * 2 * (1 + 4 + 4 + 1 + 1 + 1 + 1 + 1)),
        ]
        for nm, fn, mbytes in cases:
            if flt and flt not in nm:
                continue
            us = timeit(fn)
            out.append(f"{name} {nm:14s} {us:8.1f} us   {mbytes:7.1f} MB algorithmic   {mbytes / us:5.2f} TB/s")
    print("\n".join(out))


if __name__ == "__main__":
    main()
