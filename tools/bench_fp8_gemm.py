"""fp8 vs bf16 forward GEMM on the step's shapes (dev tooling): python tools/bench_fp8_gemm.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

shapes = [(10960, 2304, 768, "enc qkv"), (10960, 768, 768, "enc proj"), (10960, 3072, 768, "enc fc1 (+gelu)"), (10960, 768, 3072, "enc fc2"),
          (32848, 512, 512, "dec q/proj"), (32848, 2048, 512, "dec fc1 (+gelu)"), (32848, 512, 2048, "dec fc2"),
          (13832, 3072, 1024, "large qkv"), (13832, 4096, 1024, "large fc1 (+gelu)"), (13832, 1024, 4096, "large fc2")]
for m, n, k, name in shapes:
    a = torch.randn(m, k, device="cuda").bfloat16()
    w = (torch.randn(n, k, device="cuda") * 0.05).bfloat16()
    bias = torch.zeros(n, device="cuda")
    gelu = "gelu" in name
    h = torch.empty(m, n, dtype=torch.bfloat16, device="cuda") if gelu else None
    a8, sa = K.quantize_fp8(a)
    w8, sw = K.quantize_fp8(w)

    def t(fn, reps=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    us16 = t(lambda: K.gemm(a, w, bias=bias, act=int(gelu), aux_out=h))
    us8 = t(lambda: K.gemm_fp8(a8, sa, w8, sw, bias=bias, act=int(gelu), aux_out=h))
    usq = t(lambda: K.quantize_fp8(a))
    fl = 2.0 * m * n * k
    print(f"{name:20s} {m}x{n}x{k}: bf16 {us16:7.1f} us ({fl / us16 / 1e6:6.0f} TF)  fp8 {us8:7.1f} us ({fl / us8 / 1e6:6.0f} TF)  quantize A {usq:6.1f} us")
