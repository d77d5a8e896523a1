"""A/B of the persistent 256x256 GEMM (csrc/gemm256.hip) against the 128x128 kernels on the shapes of the MAE step (dev tooling).
   python tools/bench_p256.py [fwd] [dgrad] [wgrad]
Interleaved rounds in one process, HIP events, median; TFLOP/s (us)."""

from __future__ import annotations

import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402
from cinema_amd.tape import _split_k  # noqa: E402

SHAPES = [
    ("4096^3", 4096, 4096, 4096), ("8192^3", 8192, 8192, 8192),
    ("enc qkv", 10960, 2304, 768), ("enc proj", 10960, 768, 768), ("enc fc1", 10960, 3072, 768), ("enc fc2", 10960, 768, 3072),
    ("dec q/proj", 32848, 512, 512), ("dec kv", 10944, 1024, 512), ("dec fc1", 32848, 2048, 512), ("dec fc2", 32848, 512, 2048),
]


def bench(fns: dict, iters: int = 8, rounds: int = 5) -> dict:
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    times: dict = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / iters * 1e-3)
    return {k: sorted(v)[len(v) // 2] for k, v in times.items()}


def fmt(flops: float, t: float) -> str:
    return f"{flops / t / 1e12:7.1f} ({t * 1e6:7.1f})"


def main() -> None:
    modes = [a for a in sys.argv[1:] if not a.startswith("-")] or ["fwd", "dgrad", "wgrad"]
    dev = "cuda"
    if "fwd" in modes or "dgrad" in modes:
        print(f"{'shape':12s} {'M':>6s} {'N':>5s} {'K':>5s} | layout | {'128x128':>17s} {'p256 split':>17s} {'p256 whole-K':>17s} {'p256 stream':>17s}")
        for name, m, n, k in SHAPES:
            x = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
            w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
            dy = (torch.randn(m, n, device=dev) * 0.5).to(torch.bfloat16)
            bias = torch.randn(n, device=dev)
            flops = 2.0 * m * n * k
            if "fwd" in modes:
                y = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
                r = bench({"old": lambda: K.gemm(x, w, bias=bias, out=y), "split": lambda: K.gemm(x, w, bias=bias, out=y, p256=0, split_k=0),
                           "whole": lambda: K.gemm(x, w, bias=bias, out=y, p256=0, split_k=1), "stream": lambda: K.gemm(x, w, bias=bias, out=y, p256=1)})
                print(f"{name:12s} {m:6d} {n:5d} {k:5d} | fwd    | " + " ".join(fmt(flops, r[c]) for c in ("old", "split", "whole", "stream")), flush=True)
            if "dgrad" in modes:
                dx = torch.empty(m, k, dtype=torch.bfloat16, device=dev)
                kw = dict(a_kmajor=True, b_kmajor=False, out=dx)
                r = bench({"old": lambda: K.gemm(dy, w, **kw), "split": lambda: K.gemm(dy, w, p256=0, split_k=0, **kw),
                           "whole": lambda: K.gemm(dy, w, p256=0, split_k=1, **kw), "stream": lambda: K.gemm(dy, w, p256=1, **kw)})
                print(f"{name:12s} {m:6d} {n:5d} {k:5d} | dgrad  | " + " ".join(fmt(flops, r[c]) for c in ("old", "split", "whole", "stream")), flush=True)
    if "wgrad" in modes:
        blocks = {
            "enc block": [(10960, 2304, 768), (10960, 768, 768), (10960, 3072, 768), (10960, 768, 3072)],
            "dec block": [(32848, 512, 512), (10944, 1024, 512), (32848, 512, 512), (32848, 2048, 512), (32848, 512, 2048)],
            "dec no kv": [(32848, 512, 512), (32848, 512, 512), (32848, 2048, 512), (32848, 512, 2048)],
            "shared kv": [(10944, 8192, 512)],
            "large enc": [(13824, 3072, 1024), (13824, 1024, 1024), (13824, 4096, 1024), (13824, 1024, 4096)], "large mlp": [(13824, 4096, 1024), (13824, 1024, 4096)],
            "4096^3": [(4096, 4096, 4096)], "8192^3": [(8192, 8192, 8192)],
        }
        print(f"{'block':12s} | {'128x128 split-K + reduce':>26s} {'128x128 grouped whole-K':>26s} {'p256 grouped':>22s} {'p256 e4m3 operands':>22s}")
        for name, gs in blocks.items():
            probs = []
            for rows, n, k in gs:
                dy = (torch.randn(rows, n, device=dev) * 0.5).to(torch.bfloat16)
                x = (torch.randn(rows, k, device=dev) * 0.5).to(torch.bfloat16)
                probs.append((dy, x, torch.zeros(n, k, dtype=torch.float32, device=dev), torch.zeros(n, dtype=torch.float32, device=dev)))
            flops = sum(2.0 * r * n * k for r, n, k in gs)

            def old() -> None:
                for dy, x, dst, rs in probs:
                    K.gemm(dy, x, a_kmajor=False, b_kmajor=False, out=dst, accumulate=True, split_k=_split_k(dy.shape[0], dy.shape[1], x.shape[1]), a_rowsum=rs)

            q8 = [(*K.quantize_fp8(dy), *K.quantize_fp8(x), dst) for dy, x, dst, _ in probs]
            fns = {"old": old, "p256": lambda: K.gemm_wgrad_grouped(probs, p256=True), "fp8": lambda: K.gemm_fp8_wgrad_grouped(q8)}
            if len({r for r, _, _ in gs}) == 1:
                fns["grouped"] = lambda: K.gemm_wgrad_grouped(probs)
            r = bench(fns, iters=4)
            print(f"{name:12s} | {fmt(flops, r['old']):>26s} {(fmt(flops, r['grouped']) if 'grouped' in r else '-'):>26s} {fmt(flops, r['p256']):>22s} {fmt(flops, r['fp8']):>22s}", flush=True)


if __name__ == "__main__":
    main()
