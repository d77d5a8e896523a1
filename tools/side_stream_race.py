"""Race hunt for the weight-gradient streams: the same forward + backward N times (identical inputs and masks, gradients zeroed in between); every parameter whose
gradient differs from the first run's by more than rounding noise is reported (float atomics give ~1e-6 differences; a missing / torn contribution gives percent).
   CINEMA_SIDE_STREAMS=2 python tools/side_stream_race.py [iterations] [n_blocks]"""
import math
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
import cinema_oracle as O  # noqa: E402, N812  (mask recipe only)
from cinema_amd import CineMA  # noqa: E402
from cinema_amd.optim import FlatModel  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
views = ["sax", "lax_2c", "lax_3c"]
kw = dict(image_size_dict={v: (64, 64, 8) if v == "sax" else (64, 64) for v in views}, in_chans_dict=dict.fromkeys(views, 1),
          enc_patch_size_dict={v: (4, 4, 1) if v == "sax" else (4, 4) for v in views}, enc_scale_factor_dict={v: (2, 2, 1) if v == "sax" else (2, 2) for v in views},
          enc_conv_chans=[64, 128], enc_conv_n_blocks=nb, enc_embed_dim=256, enc_depth=2, enc_n_heads=4, dec_embed_dim=128, dec_depth=2, dec_n_heads=4)
cfg = O.MAEConfig(**kw)
gen = torch.Generator().manual_seed(3)
images = {v: torch.rand(4, 1, *s, generator=gen).cuda() for v, s in kw["image_size_dict"].items()}
masks = {v: O.random_patch_mask(4, math.prod(cfg.grid_size(v)), 0.75, gen).cuda() for v in images}
torch.manual_seed(0)
model = CineMA(**kw).cuda()
flat = FlatModel(model, 0.05)
names = {flat.offsets[id(p)][0]: k for k, p in model.named_parameters() if id(p) in flat.offsets}
ref, bad = None, 0
for it in range(n):
    flat.zero_grad()
    loss, _, _, _ = model(images, 0.75, enc_mask_dict=masks)
    loss.backward()
    torch.cuda.synchronize()
    g = flat.flat_grad.clone()
    if ref is None:
        ref = g
        continue
    diff = (g - ref).abs()
    if float(diff.max()) > 1e-3 * float(ref.abs().max()):
        bad += 1
        worst = []
        for p in model.parameters():
            a, b = flat.offsets[id(p)]
            d, r = float(diff[a:b].max()), float(ref[a:b].abs().max())
            if d > 1e-3 * max(r, 1e-12):
                worst.append((names[a], round(d / max(r, 1e-12), 4)))
        print(f"iteration {it}: {len(worst)} tensors differ: {worst[:8]}", flush=True)
print("RACE HUNT", "clean" if bad == 0 else f"{bad} of {n - 1} iterations differ")
