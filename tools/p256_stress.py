"""Stress of the in-launch reduction of the persistent GEMM: the weight gradients of an encoder and a decoder block N times on the same operands, every result
compared BIT for bit with the first (a stale or torn partial tile would show up as a difference), counters checked at the end (dev tooling)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cinema_amd import hip as K  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
BLOCKS = {
    "enc": [(10960, 2304, 768), (10960, 768, 768), (10960, 3072, 768), (10960, 768, 3072)],
    "dec": [(32848, 512, 512), (10944, 1024, 512), (32848, 512, 512), (32848, 2048, 512), (32848, 512, 2048)],
}
bad = 0
for name, gs in BLOCKS.items():
    ops = []
    for rows, nn, kk in gs:
        ops.append(((torch.randn(rows, nn, device="cuda") * 0.5).to(torch.bfloat16), (torch.randn(rows, kk, device="cuda") * 0.5).to(torch.bfloat16)))
    ref = None
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, device="cuda")
    for it in range(n):
        probs = [(dy, x, torch.zeros(dy.shape[1], x.shape[1], device="cuda"), torch.zeros(dy.shape[1], device="cuda")) for dy, x in ops]
        if it % 3 == 0:  # other traffic beside the launch: a streaming kernel on a second stream
            with torch.cuda.stream(side):
                junk.mul_(1.0001)
        K.gemm_wgrad_grouped(probs, p256=True)
        out = [p[2] for p in probs]
        if ref is None:
            ref = [o.clone() for o in out]
        elif not all(torch.equal(a, b) for a, b in zip(out, ref)):
            bad += 1
            print(name, "iteration", it, "differs:", [float((a - b).abs().max()) for a, b in zip(out, ref)], flush=True)
    torch.cuda.synchronize()
    print(name, n, "launches, mismatches so far", bad, flush=True)
ws = K._p256_workspace(torch.device("cuda", torch.cuda.current_device()))
left = int(ws[:16384].view(torch.int32).abs().sum())
print("STRESS", "OK" if bad == 0 and left == 0 else "FAILED", {"mismatches": bad, "counter_sum": left})
sys.exit(0 if bad == 0 and left == 0 else 1)
