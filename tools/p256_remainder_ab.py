import os, sys, torch
sys.path.insert(0, ".")
from cinema_amd import hip as K
from tools.bench_p256_loop import bench
dev = "cuda"
blocks = {
    "enc block": [(10960, 2304, 768), (10960, 768, 768), (10960, 3072, 768), (10960, 768, 3072)],
    "dec no kv": [(32848, 512, 512), (32848, 512, 512), (32848, 2048, 512), (32848, 512, 2048)],
    "large enc": [(13824, 3072, 1024), (13824, 1024, 1024), (13824, 4096, 1024), (13824, 1024, 4096)],
}
for name, gs in blocks.items():
    probs = []
    for rows, n, k in gs:
        dy = (torch.randn(rows, n, device=dev) * 0.5).to(torch.bfloat16)
        x = (torch.randn(rows, k, device=dev) * 0.5).to(torch.bfloat16)
        probs.append((dy, x, torch.zeros(n, k, dtype=torch.float32, device=dev), torch.zeros(n, dtype=torch.float32, device=dev)))
    flops = sum(2.0 * r * n * k for r, n, k in gs)
    def mk(loop, rem):
        def run():
            os.environ["CINEMA_P256_LOOP"] = str(loop); os.environ["CINEMA_P256_REMAINDER"] = str(rem)
            K.gemm_wgrad_grouped(probs, p256=True)
        return run
    r = bench({(l, m): mk(l, m) for l in (1, 2) for m in (0, 1)})
    print(f"{name:10s} | " + "  ".join(f"loop {l} rem {m}: {flops / r[(l, m)] / 1e12:6.1f} TF ({r[(l, m)] * 1e6:6.1f} us)" for l in (1, 2) for m in (0, 1)), flush=True)
