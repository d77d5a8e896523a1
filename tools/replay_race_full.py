"""Race hunt at the REAL config-2 shape in the product's own mode: one recorded step (cinema_amd/replay.py) replayed N times on identical inputs and identical masks
(draw_masks pinned), gradients zeroed in between; every parameter whose gradient differs from the first replay's by more than atomics noise is reported.  With
kernels of the real durations on three / four streams a lost update (two accumulating launches on one buffer at the same time) shows as a per-cent difference.
   [CINEMA_SIDE_STREAMS=2] python tools/replay_race_full.py [replays] [--break]      --break: self-test, re-introduces the shared-destination bug"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from cinema_amd import CineMA  # noqa: E402
from cinema_amd import tape as T  # noqa: E402
from cinema_amd.optim import FlatModel  # noqa: E402
from cinema_amd.replay import RecordedStep  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 40
if "--break" in sys.argv:  # self-test of the hunter: forget which stream a destination was written from (the round-5 bug: dec_linear's four accumulating launches)
    class _Forgetful(dict):
        def __contains__(self, key) -> bool:  # noqa: ANN001
            return False
    T._DST_STREAM = _Forgetful()  # noqa: SLF001
kw = bench.base_kwargs("base")
torch.manual_seed(0)
model = CineMA(**kw).to("cuda")
flat = FlatModel(model, 0.05)
batch = bench.synthetic_batch(kw, 16, 1, "cuda")
fixed = model.draw_masks(batch, 0.75)
model.draw_masks = lambda images, ratio: fixed  # noqa: ARG005  (every replay sees the same masks)
rec = RecordedStep(model, batch, 0.75)
names = {id(p): k for k, p in model.named_parameters()}
ref, worst, bad = None, 0.0, {}
for it in range(n):
    flat.zero_grad()
    torch.cuda.synchronize()
    rec.run(batch)
    torch.cuda.synchronize()
    g = flat.flat_grad.clone()
    if ref is None:
        ref = g
        continue
    for p in flat.params:
        a, b = flat.offsets[id(p)]
        d = float((g[a:b] - ref[a:b]).norm() / ref[a:b].norm().clamp_min(1e-20))
        worst = max(worst, d)
        if d > 1e-3:
            bad[names[id(p)]] = max(bad.get(names[id(p)], 0.0), d)
print(f"REPLAY RACE HUNT side_streams {T.SIDE_STREAMS} replays {n} grad norm {float(ref.norm()):.4f} worst per-tensor rel diff {worst:.2e}", "clean" if not bad else f"BAD {bad}")
