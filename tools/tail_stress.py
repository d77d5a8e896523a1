"""Stress of the split tail finished inside the launch (reduce-scatter over the k-slices, csrc/gemm.hip tail_finish_in_launch): the step's tail-carrying GEMM
shapes N times on the same operands, every result compared BIT for bit with the first launch and with the fix-up form, beside a second stream that keeps the
chip busy with persistent 256x256 weight-gradient launches (they occupy whole CUs, so tail slices of one tile start at different times - the situation the
bounded wait is for); counters and the error word checked at the end (dev tooling)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = "cuda"
SHAPES = [(10960, 768, 768), (10960, 2304, 768), (10960, 3072, 768), (10960, 768, 3072), (32848, 512, 2048)]
side = torch.cuda.Stream()
wg = [((torch.randn(10960, nn, device=dev) * 0.5).to(torch.bfloat16), (torch.randn(10960, kk, device=dev) * 0.5).to(torch.bfloat16)) for nn, kk in ((2304, 768), (3072, 768))]
bad = 0
for m, nn, kk in SHAPES:
    x = (torch.randn(m, kk, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(nn, kk, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(nn, device=dev)
    res = torch.randn(m, nn, device=dev)
    K.TAIL_IN_LAUNCH = False
    ref = [K.gemm(x, w, bias=bias).clone(), K.gemm(x, w, bias=bias, residual=res, out_dtype=torch.float32).clone(), K.gemm(x, w, bias=bias, act=1).clone()]
    K.TAIL_IN_LAUNCH = True
    for it in range(n):
        if it % 2 == 0:
            with torch.cuda.stream(side), K.on_stream(side.cuda_stream):
                K.gemm_wgrad_grouped([(dy, xx, torch.zeros(dy.shape[1], xx.shape[1], device=dev), None) for dy, xx in wg], p256=True)
        out = [K.gemm(x, w, bias=bias), K.gemm(x, w, bias=bias, residual=res, out_dtype=torch.float32), K.gemm(x, w, bias=bias, act=1)]
        if not all(torch.equal(a, b) for a, b in zip(out, ref)):
            bad += 1
            if bad < 5:
                print((m, nn, kk), "iteration", it, "differs:", [float((a.float() - b.float()).abs().max()) for a, b in zip(out, ref)], flush=True)
    torch.cuda.synchronize()
    print((m, nn, kk), n, "x 3 launches, mismatches so far", bad, flush=True)
cnt = K._tail_counters(torch.device(dev, torch.cuda.current_device()))  # noqa: SLF001
left = int(cnt.abs().sum())
print("TAIL STRESS", "OK" if bad == 0 and left == 0 else "FAILED", {"mismatches": bad, "counter_and_error_word_sum": left})
sys.exit(0 if bad == 0 and left == 0 else 1)
