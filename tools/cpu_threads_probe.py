"""How fast is the CPU oracle on this host for different intra-op thread counts? (dev tooling)"""
import math, os, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import bench
import cinema_oracle as O
kw = bench.base_kwargs("base")
cfg = O.MAEConfig(**kw)
p = O.init_params(cfg, 0)
for nt in [int(x) for x in sys.argv[1:]] or [32]:
    torch.set_num_threads(nt)
    tr = O.Trainer(p, cfg)
    gen = torch.Generator().manual_seed(1)
    images = {v: torch.rand(2, 1, *s, generator=gen) for v, s in kw["image_size_dict"].items()}
    masks = {v: O.random_patch_mask(2, math.prod(cfg.grid_size(v)), 0.75, gen) for v in images}
    t0 = time.perf_counter(); tr.step(images, masks); t1 = time.perf_counter(); tr.step(images, masks); t2 = time.perf_counter()
    print(f"threads {nt:3d}: warm-up step {t1-t0:6.1f} s, step {t2-t1:6.1f} s -> {2/(t2-t1):.3f} samples/s", flush=True)
