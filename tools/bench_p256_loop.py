"""A/B of the main-loop forms of the persistent 256x256 GEMM (csrc/gemm256.hip; CINEMA_P256_LOOP is read per call) on the step's weight-gradient groups and
on the forward / data-gradient layouts (dev tooling).
   python tools/bench_p256_loop.py [forms, default 1,2]
Interleaved rounds in one process, HIP events, median; TFLOP/s (us).  Also checks that the forms agree bit for bit (same summation order)."""

from __future__ import annotations

import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402


def bench(fns: dict, iters: int = 6, rounds: int = 7) -> dict:
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


def with_form(form: int, fn):
    def run():
        os.environ["CINEMA_P256_LOOP"] = str(form)
        fn()
    return run


def main() -> None:
    forms = [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "2"])]
    dev = "cuda"
    blocks = {
        "enc block": [(10960, 2304, 768), (10960, 768, 768), (10960, 3072, 768), (10960, 768, 3072)],
        "dec no kv": [(32848, 512, 512), (32848, 512, 512), (32848, 2048, 512), (32848, 512, 2048)],
        "shared kv": [(10944, 8192, 512)],
        "large enc": [(13824, 3072, 1024), (13824, 1024, 1024), (13824, 4096, 1024), (13824, 1024, 4096)],
        "4096^3": [(4096, 4096, 4096)], "8192^3": [(8192, 8192, 8192)],
    }
    print("weight-gradient groups (reduction-strided operands) | " + " ".join(f"{'form ' + str(f):>20s}" for f in forms))
    for name, gs in blocks.items():
        probs = []
        for rows, n, k in gs:
            dy = (torch.randn(rows, n, device=dev) * 0.5).to(torch.bfloat16)
            x = (torch.randn(rows, k, device=dev) * 0.5).to(torch.bfloat16)
            probs.append((dy, x, torch.zeros(n, k, dtype=torch.float32, device=dev), torch.zeros(n, dtype=torch.float32, device=dev)))
        flops = sum(2.0 * r * n * k for r, n, k in gs)
        outs = []
        for f in forms:
            for _, _, dst, rs in probs:
                dst.zero_(); rs.zero_()
            with_form(f, lambda: K.gemm_wgrad_grouped(probs, p256=True))()
            torch.cuda.synchronize()
            outs.append([dst.clone() for _, _, dst, _ in probs])
        same = all(torch.equal(a, b) for o in outs[1:] for a, b in zip(outs[0], o))
        r = bench({f: with_form(f, lambda: K.gemm_wgrad_grouped(probs, p256=True)) for f in forms})
        print(f"{name:12s} | " + " ".join(f"{flops / r[f] / 1e12:9.1f} ({r[f] * 1e6:7.1f})" for f in forms) + f" | identical {same}", flush=True)
    print("forward / data-gradient layouts, whole-K tiles | " + " ".join(f"{'form ' + str(f):>20s}" for f in forms))
    for name, m, n, k in [("4096^3", 4096, 4096, 4096), ("8192^3", 8192, 8192, 8192), ("enc fc1", 10752, 3072, 768), ("dec fc1", 32768, 2048, 512)]:
        x = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        dy = (torch.randn(m, n, device=dev) * 0.5).to(torch.bfloat16)
        y = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        dx = torch.empty(m, k, dtype=torch.bfloat16, device=dev)
        flops = 2.0 * m * n * k
        for lay, fn, out in (("fwd", lambda: K.gemm(x, w, out=y, p256=0, split_k=1), y),
                             ("dgrad", lambda: K.gemm(dy, w, a_kmajor=True, b_kmajor=False, out=dx, p256=0, split_k=1), dx)):
            outs = []
            for f in forms:
                with_form(f, fn)()
                torch.cuda.synchronize()
                outs.append(out.clone())
            same = all(torch.equal(outs[0], o) for o in outs[1:])
            r = bench({f: with_form(f, fn) for f in forms})
            print(f"{name:8s} {lay:5s} | " + " ".join(f"{flops / r[f] / 1e12:9.1f} ({r[f] * 1e6:7.1f})" for f in forms) + f" | identical {same}", flush=True)


if __name__ == "__main__":
    main()
