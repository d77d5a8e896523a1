"""Golden vectors for the segmentation evaluation path, generated from the upstream reference (runs ONLY where
/root/reference exists):  python oracle/make_golden_seg_eval.py  ->  tests/golden/seg_eval.safetensors

What is captured: ``get_patch_grid`` grids, ``patch_grid_sample`` + ``aggregate_patches`` on random data, and ``segmentation_forward`` (sliding
window) of the reference ConvUNetR mini model (weights of tests/golden/convunetr_mini.safetensors) on an image larger than its patch size.
monai is not installed: ``cinema/segmentation/train.py`` imports it at module level, so a placeholder module whose members raise when CALLED
lets the module import; nothing captured here touches monai (``segmentation_forward`` uses only torch and ``cinema/transform.py``).
"""

from __future__ import annotations

import json
import sys
import types
from pathlib import Path

import numpy as np
import torch
from safetensors.torch import load_file, save_file

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import ref_shim  # noqa: E402

ref_shim.install()


def _absent(name: str):  # noqa: ANN202
    def fail(*a, **k):  # noqa: ANN002, ANN003, ANN202
        raise RuntimeError(f"monai.{name} is not installed here")
    return fail


for mod, names in (("monai", ()), ("monai.losses", ("DiceLoss",)), ("monai.metrics", ("compute_dice", "compute_hausdorff_distance", "compute_iou")),
                   ("monai.networks", ()), ("monai.networks.utils", ("one_hot",))):
    m = types.ModuleType(mod)
    for n in names:
        setattr(m, n, _absent(n))
    sys.modules.setdefault(mod, m)

from cinema.segmentation.convunetr import ConvUNetR  # noqa: E402
from cinema.segmentation.train import segmentation_forward  # noqa: E402
from cinema.transform import aggregate_patches, get_patch_grid, patch_grid_sample  # noqa: E402

OUT = HERE.parent / "tests" / "golden"


def main() -> None:
    torch.set_num_threads(8)
    t = {}
    cases = [((6, 11), (3, 5), (1, 3)), ((8, 10, 6), (4, 5, 6), (2, 1, 4)), ((96, 80, 6), (64, 64, 4), (32, 32, 2)), ((192, 128, 128), (128, 128, 128), (64, 0, 0))]
    for i, (size, patch, ov) in enumerate(cases):
        t[f"grid/{i}/args"] = torch.tensor([*size, *patch, *ov], dtype=torch.int64)
        t[f"grid/{i}/starts"] = torch.from_numpy(get_patch_grid(image_size=size, patch_size=patch, patch_overlap=ov).astype(np.int64))
    g = torch.Generator().manual_seed(11)
    x = torch.rand(3, 8, 10, 6, generator=g)
    starts = get_patch_grid((8, 10, 6), (4, 5, 6), (2, 1, 4))
    patches = patch_grid_sample(x, starts, (4, 5, 6))
    noise = torch.rand(patches.shape, generator=g)
    t["agg/patches"], t["agg/out"] = noise, aggregate_patches(noise, starts, (8, 10, 6))
    t["agg/sampled"] = patches

    meta = json.loads((OUT / "convunetr_meta.json").read_text())
    kw = meta["kwargs"]
    for key in ("image_size_dict", "enc_patch_size_dict", "enc_scale_factor_dict", "dec_patch_size_dict", "dec_scale_factor_dict"):
        kw[key] = {v: tuple(s) for v, s in kw[key].items()}
    kw["dec_chans"] = tuple(kw["dec_chans"])
    model = ConvUNetR(**kw)
    gold = load_file(str(OUT / "convunetr_mini.safetensors"))
    model.load_state_dict({k[len("param/"):]: v for k, v in gold.items() if k.startswith("param/")})
    model.eval()
    # the SAX view is larger than its (64, 64, 4) patch: 3 x 2 x 2 = 12 half-overlapping windows; the LAX view is passed whole
    images = {"sax": torch.rand(1, 1, 96, 80, 6, generator=g), "lax_4c": torch.rand(1, 1, 64, 64, generator=g)}
    out = segmentation_forward(model, images, {"sax": (64, 64, 4), "lax_4c": (64, 64)}, torch.bfloat16)
    for v in images:
        t[f"fwd/image/{v}"], t[f"fwd/logits/{v}"] = images[v], out[v].detach().float()
    # no view needs patching: the plain forward
    whole = {"sax": images["sax"][:, :, :64, :64, :4].contiguous(), "lax_4c": images["lax_4c"]}
    out2 = segmentation_forward(model, whole, {"sax": (64, 64, 4), "lax_4c": (64, 64)}, torch.bfloat16)
    t["fwd/whole_logits/sax"] = out2["sax"].detach().float()
    save_file({k: v.detach().clone().contiguous() for k, v in t.items()}, str(OUT / "seg_eval.safetensors"))
    print("wrote seg_eval.safetensors", sum(v.numel() * v.element_size() for v in t.values()) / 1e6, "MB")


if __name__ == "__main__":
    main()
