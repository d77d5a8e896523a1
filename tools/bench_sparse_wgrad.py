"""The visible-voxel depthwise weight gradient at the shapes of the MAE step (SAX stage 1 / 2, one long-axis view): pipelined kernel vs the per-token
index chase (dev tooling).   python tools/bench_sparse_wgrad.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

dev = "cuda"
CASES = [("sax stage 1", 16, (12, 12, 16), (4, 4, 1), 64), ("sax stage 2", 16, (12, 12, 16), (2, 2, 1), 128), ("lax stage 1", 16, (12, 12), (4, 4), 64)]
for name, b, tok_grid, block, c in CASES:
    T = 1
    for g in tok_grid:
        T *= g
    gen = torch.Generator().manual_seed(0)
    keep = torch.cat([torch.randperm(T, generator=gen)[:T // 4].sort().values + i * T for i in range(b)]).to(torch.int32).to(dev)
    rank = torch.full((b * T,), -1, dtype=torch.int32, device=dev)
    rank[keep.long()] = torch.arange(keep.numel(), dtype=torch.int32, device=dev)
    bv = 1
    for v in block:
        bv *= v
    pos = torch.arange(bv, dtype=torch.int32, device=dev)
    geom = K.sparse_geom(b, tok_grid, block, keep, rank, pos)
    rows = keep.numel() * bv
    x = (torch.randn(rows, c, device=dev) * 0.5).to(torch.bfloat16)
    dy = (torch.randn(rows, c, device=dev) * 0.5).to(torch.bfloat16)
    wshape = (c, 1) + (5,) * len(tok_grid)
    dw, db = torch.zeros(wshape, device=dev), torch.zeros(c, device=dev)
    out = {}
    for pipe in (True, False):
        K.SPARSE_WGRAD_PIPE = pipe
        for _ in range(3):
            K.sparse_dwconv_bwd_weight(x, dy, wshape, dw, db, geom)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            K.sparse_dwconv_bwd_weight(x, dy, wshape, dw, db, geom)
        e1.record()
        torch.cuda.synchronize()
        out[pipe] = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:12s} rows {rows:7d} c {c:4d}: pipelined {out[True]:7.1f} us   index chase {out[False]:7.1f} us (kernel + slab reduce)", flush=True)
