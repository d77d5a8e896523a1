"""Split-tail A/B on the step's GEMM shapes (dev tooling): fix-up launch vs the tail finished inside the launch (reduce-scatter over the k-slices).
   python tools/tail_ab.py"""
from __future__ import annotations

import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

SHAPES = [("enc qkv", 10960, 2304, 768), ("enc proj", 10960, 768, 768), ("enc fc1", 10960, 3072, 768), ("enc fc2", 10960, 768, 3072),
          ("dec q/proj", 32848, 512, 512), ("dec kv", 10944, 1024, 512), ("dec fc1", 32848, 2048, 512), ("dec fc2", 32848, 512, 2048)]


def timeit(fns: dict, iters: int = 20, rounds: int = 5) -> dict:
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / iters * 1e3)
    return {k: sorted(v)[len(v) // 2] for k, v in res.items()}


def main() -> None:
    dev = "cuda"
    print(f"{'shape':12s} {'M':>6s} {'N':>5s} {'K':>5s} | fwd bf16: fixup / in-launch us | fwd f32+res | fwd gelu+deriv | dgrad | dgrad x gelu'")
    for name, m, n, k in SHAPES:
        x = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        dy = (torch.randn(m, n, device=dev) * 0.5).to(torch.bfloat16)
        bias = torch.randn(n, device=dev)
        res = torch.randn(m, n, device=dev)
        gin = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
        y16, y32, aux = torch.empty(m, n, dtype=torch.bfloat16, device=dev), torch.empty(m, n, device=dev), torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        dx = torch.empty(m, k, dtype=torch.bfloat16, device=dev)

        def mk(fn):  # noqa: ANN001, ANN202
            def off():  # noqa: ANN202
                K.TAIL_IN_LAUNCH = False
                fn()

            def on():  # noqa: ANN202
                K.TAIL_IN_LAUNCH = True
                fn()
            return {"fix": off, "in": on}

        cols = []
        for fn in (lambda: K.gemm(x, w, bias=bias, out=y16), lambda: K.gemm(x, w, bias=bias, residual=res, out=y32),
                   lambda: K.gemm(x, w, bias=bias, act=1, aux_out=aux, gelu_deriv=True, out=y16),
                   lambda: K.gemm(dy, w, a_kmajor=True, b_kmajor=False, out=dx), lambda: K.gemm(dy, w, a_kmajor=True, b_kmajor=False, gelu_in=gin, gelu_deriv=True, out=dx)):
            t = timeit(mk(fn))
            cols.append(f"{t['fix']:6.1f} / {t['in']:6.1f}")
        print(f"{name:12s} {m:6d} {n:5d} {k:5d} | " + " | ".join(cols), flush=True)
    K.TAIL_IN_LAUNCH = False
    err = int(K._tail_counters(torch.device(dev, 0))[2047])  # noqa: SLF001
    print("tail error word:", err)


if __name__ == "__main__":
    main()
