"""GEMM micro-benchmark for the shapes of the MAE step (dev tooling).  Usage on the GPU box:
   python tools/bench_gemm.py [fwd|dgrad|wgrad|all] [--split S]
Prints TFLOP/s per (shape, layout) with HIP-event timing (median of interleaved rounds)."""

from __future__ import annotations

import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

# (M tokens, N out, K in) of Y = X W^T
SHAPES = [
    ("4096^3", 4096, 4096, 4096),
    ("enc qkv", 10960, 2304, 768), ("enc proj", 10960, 768, 768), ("enc fc1", 10960, 3072, 768), ("enc fc2", 10960, 768, 3072),
    ("dec q/proj", 32848, 512, 512), ("dec kv", 10944, 1024, 512), ("dec fc1", 32848, 2048, 512), ("dec fc2", 32848, 512, 2048),
    ("X dec kv fused", 10944, 8192, 512), ("X dec kv one", 10944, 1024, 512),
    ("F enc qkv", 10752, 2304, 768), ("F enc proj", 10752, 768, 768), ("F enc fc1", 10752, 3072, 768), ("F enc fc2", 10752, 768, 3072),
    ("F dec proj", 32768, 512, 512), ("F dec fc1", 32768, 2048, 512), ("F dec fc2", 32768, 512, 2048),
    ("stem1 1x1", 589824, 64, 64), ("stem1 fc1", 589824, 256, 64), ("stem1 fc2", 589824, 64, 256),
    ("stem2 1x1", 147456, 128, 128), ("stem2 fc1", 147456, 512, 128), ("stem2 fc2", 147456, 128, 512), ("stem2 conv", 147456, 128, 256),
]


def timeit(fn, iters: int = 10, rounds: int = 3) -> float:  # noqa: ANN001
    fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / iters * 1e-3)
    return sorted(best)[len(best) // 2]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", nargs="?", default="all")
    ap.add_argument("--split", type=int, default=0)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = "cuda"
    print(f"{'shape':12s} {'M':>7s} {'N':>5s} {'K':>5s} | " + " ".join(f"{m:>22s}" for m in ("fwd TF (us)", "dgrad TF (us)", "wgrad TF (us) [split]")))
    for name, m, n, k in SHAPES:
        if args.only and args.only not in name:
            continue
        x = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        dy = (torch.randn(m, n, device=dev) * 0.5).to(torch.bfloat16)
        bias = torch.randn(n, device=dev)
        flops = 2.0 * m * n * k
        out = []
        if args.mode in ("fwd", "all"):
            y = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
            t = timeit(lambda: K.gemm(x, w, bias=bias, out=y))
            out.append(f"{flops / t / 1e12:8.1f} ({t * 1e6:8.1f})")
        else:
            out.append(" " * 19)
        if args.mode in ("dgrad", "all"):
            dx = torch.empty(m, k, dtype=torch.bfloat16, device=dev)
            t = timeit(lambda: K.gemm(dy, w, a_kmajor=True, b_kmajor=False, out=dx))
            out.append(f"{flops / t / 1e12:8.1f} ({t * 1e6:8.1f})")
        else:
            out.append(" " * 19)
        if args.mode in ("wgrad", "all"):
            from cinema_amd.tape import _split_k

            sp = args.split or _split_k(m, n, k)
            dw = torch.zeros(n, k, dtype=torch.float32, device=dev)
            t = timeit(lambda: K.gemm(dy, x, a_kmajor=False, b_kmajor=False, out=dw, accumulate=True, split_k=sp))
            out.append(f"{flops / t / 1e12:8.1f} ({t * 1e6:8.1f}) [{sp}]")
        print(f"{name:12s} {m:7d} {n:5d} {k:5d} | " + "   ".join(out), flush=True)


if __name__ == "__main__":
    main()
