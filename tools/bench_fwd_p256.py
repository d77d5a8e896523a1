"""Forward GEMMs of the step with their real epilogues: the 128x128 kernels against the persistent 256x256 kernel (split and stream schedules), alone on the chip as in
the forward pass (no weight-gradient stream there).  Dev tooling.   python tools/bench_fwd_p256.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
# (name, m, n, k, epilogue)   epilogue: "bf16" plain, "gelu" (act + GELU' aux), "res" (fp32 out + residual)
SHAPES = (("enc qkv", 10960, 2304, 768, "bf16"), ("enc proj", 10960, 768, 768, "res"), ("enc fc1", 10960, 3072, 768, "gelu"), ("enc fc2", 10960, 768, 3072, "res"),
          ("dec q", 32848, 512, 512, "bf16"), ("dec proj", 32848, 512, 512, "res"), ("dec fc1", 32848, 2048, 512, "gelu"), ("dec fc2", 32848, 512, 2048, "res"),
          ("dec kv x8", 10944, 8192, 512, "bf16"))
tot = {}
for name, m, n, k, epi in SHAPES:
    x = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(n, device=dev)
    res = torch.randn(m, n, device=dev) if epi == "res" else None
    aux = torch.empty(m, n, dtype=torch.bfloat16, device=dev) if epi == "gelu" else None
    out = torch.empty(m, n, dtype=torch.float32 if epi == "res" else torch.bfloat16, device=dev)

    def run(p256):  # noqa: ANN001, ANN202
        return K.gemm(x, w, bias=bias, residual=res, out=out, act=1 if epi == "gelu" else 0, aux_out=aux, gelu_deriv=epi == "gelu", p256=p256)

    line = f"{name:10s} {m:6d} {n:5d} {k:5d} {epi:5s}|"
    ref = None
    for label, p in (("128x128", None), ("p256 split", 0), ("p256 stream", 1)):
        try:
            t = timeit(lambda: run(p), iters=20)
        except Exception as e:  # noqa: BLE001
            line += f" {label}: {type(e).__name__}"
            continue
        y = out.float().clone()
        if ref is None:
            ref = y
        err = float((y - ref).abs().max() / ref.abs().max())
        tot[label] = tot.get(label, 0.0) + t * (12 if name.startswith("enc") else (8 if name != "dec kv x8" else 1))
        line += f" {label} {t * 1e6:7.1f} us {2.0 * m * n * k / t / 1e12:5.0f} TF (err {err:.0e}) |"
    print(line, flush=True)
print("forward GEMM time per step (12 encoder / 8 decoder blocks):", {k: round(v * 1e3, 3) for k, v in tot.items()}, "ms")
