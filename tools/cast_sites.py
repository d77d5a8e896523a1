"""Which call sites launch the fp32 <-> bf16 cast kernel (and row_copy_multi) in one MAE step, with element counts (dev tooling)."""
import collections
import sys
import traceback
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from cinema_amd import CineMA  # noqa: E402
from cinema_amd import hip as K  # noqa: E402
from cinema_amd.optim import TrainStep  # noqa: E402

SEG = "--seg" in sys.argv   # BASELINE config 4 (ConvUNetR fine-tuning step) instead of the MAE step
if SEG:
    from cinema_amd.segmentation.convunetr import ConvUNetR
    from cinema_amd.segmentation.train import SegTrainStep

    kw = bench.seg_kwargs("base", (256, 256, 12))
    torch.manual_seed(0)
    model = ConvUNetR(**kw).to("cuda").train()
    seg_step = SegTrainStep(model, ["sax"], lr=1e-4, layer_decay=0.75)
    image = torch.rand(4, 1, 256, 256, 12)
    seg_batch = {"sax_image": image.cuda(), "sax_label": torch.clamp((image * 4).long(), 0, 3).to(torch.int8).cuda()}

    def step(_b, _r):  # noqa: ANN001, ANN202
        return seg_step(seg_batch)

    batch = None
else:
    kw = bench.base_kwargs("base")
    torch.manual_seed(0)
    model = CineMA(**kw).to("cuda")
    step = TrainStep(model)
    batch = bench.synthetic_batch(kw, 16, 1, "cuda")
for _ in range(3):
    step(batch, 0.75)
sites: dict = collections.defaultdict(lambda: [0, 0])
names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["cast", "row_copy_multi", "full", "patch_weight_rows", "patch_weight_grad_accumulate"]
for name in names:
    orig = getattr(K, name)

    def wrap(*a, _orig=orig, _name=name, **k):
        st = [f for f in traceback.extract_stack()[:-1] if "cinema_amd" in f.filename and "/hip/" not in f.filename][-2:]
        key = _name + " <- " + " <- ".join(f"{Path(f.filename).name}:{f.lineno} {f.name}" for f in reversed(st))
        t = next((x for x in a if isinstance(x, torch.Tensor)), None)
        sites[key][0] += 1
        sites[key][1] += t.numel() if t is not None else 0
        return _orig(*a, **k)

    setattr(K, name, wrap)
step(batch, 0.75)
torch.cuda.synchronize()
for k, (n, el) in sorted(sites.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:4d} x  {el / max(n, 1) / 1e6:8.2f} M elements each   {k}")
