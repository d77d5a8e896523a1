"""Golden vectors for the ConvViT fine-tuning heads, generated from the upstream reference (runs ONLY where /root/reference
exists):  python oracle/make_golden_heads.py  ->  tests/golden/convvit_heads.safetensors

Captured with the reference ConvViT mini model (weights of tests/golden/convvit_mini.safetensors): ``classification_loss`` (label smoothing 0.1) and
``regression_loss`` on a batch of 2 with their metric values and the gradients of a few parameters, and ``classification_forward`` /
``regression_forward`` with an over-sized SAX view (half-overlapping patches) and with every view at its patch size.  Data only.
torchvision is not installed: ``cinema/classification/train.py`` imports the ResNet baselines (``cinema/resnet.py`` -> torchvision) at module level, so a
placeholder ``cinema.resnet`` whose members raise when CALLED lets the module import; nothing captured here touches it.
"""

from __future__ import annotations

import json
import sys
import types
from pathlib import Path

import torch
from safetensors.torch import load_file, save_file

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import ref_shim  # noqa: E402

ref_shim.install()


def _absent(name: str):  # noqa: ANN202
    def fail(*a, **k):  # noqa: ANN002, ANN003, ANN202
        raise RuntimeError(f"cinema.resnet.{name} needs torchvision, which is not installed here")
    return fail


_resnet = types.ModuleType("cinema.resnet")
_resnet.get_resnet2d, _resnet.get_resnet3d = _absent("get_resnet2d"), _absent("get_resnet3d")
sys.modules.setdefault("cinema.resnet", _resnet)

from cinema.classification.train import classification_forward, classification_loss  # noqa: E402
from cinema.convvit import ConvViT  # noqa: E402
from cinema.regression.train import regression_forward, regression_loss  # noqa: E402

OUT = HERE.parent / "tests" / "golden"
GRADS = ("pred_head_dict.sax.weight", "pred_head_dict.cls.bias", "encoder.blocks.1.mlp.fc1.weight", "enc_down_dict.sax.conv_blocks.0.patch_embed.conv.weight",
         "enc_down_dict.lax_2c.linear.bias", "encoder.cls_token")


def main() -> None:
    torch.set_num_threads(8)
    kw = json.loads((OUT / "convvit_meta.json").read_text())["kwargs"]
    for key in ("image_size_dict", "enc_patch_size_dict", "enc_scale_factor_dict"):
        kw[key] = {v: tuple(s) for v, s in kw[key].items()}
    gold = load_file(str(OUT / "convvit_mini.safetensors"))
    model = ConvViT(**kw)
    model.load_state_dict({k[len("param/"):]: v for k, v in gold.items() if k.startswith("param/")})
    model.eval()
    views, dev = ["sax", "lax_2c"], torch.device("cpu")
    named = dict(model.named_parameters())
    t: dict = {}
    batch = {f"{v}_image": gold[f"image/{v}"] for v in views}
    # classification: 3 classes, batch 2
    batch["label"] = torch.tensor([2, 0])
    model.zero_grad()
    loss, metrics = classification_loss(model, batch, views, dev, label_smoothing=0.1)
    loss.backward()
    t["cls/label"], t["cls/loss"] = batch["label"], loss.detach().reshape(1)
    t["cls/metrics"] = torch.tensor([metrics["cross_entropy"], metrics["loss"]])
    t["cls/logits"] = model({v: batch[f"{v}_image"] for v in views}).detach()
    for n in GRADS:
        t[f"cls/grad/{n}"] = named[n].grad.detach().clone()
    # regression: 3 targets per sample
    g = torch.Generator().manual_seed(5)
    batch["label"] = torch.randn(2, 3, generator=g)
    model.zero_grad()
    loss, metrics = regression_loss(model, batch, views, dev)
    loss.backward()
    t["reg/label"], t["reg/loss"] = batch["label"], loss.detach().reshape(1)
    t["reg/metrics"] = torch.tensor([metrics[k] for k in ("mse_loss", "mae_loss", "max_label", "min_label", "max_pred", "min_pred", "loss")])
    for n in GRADS:
        t[f"reg/grad/{n}"] = named[n].grad.detach().clone()
    # evaluation forward: SAX 48 x 40 x 4 against the (32, 32, 4) patch -> 2 x 2 x 1 half-overlapping patches; LAX whole
    images = {"sax": torch.rand(1, 2, 48, 40, 4, generator=g), "lax_2c": torch.rand(1, 2, 32, 32, generator=g)}
    sizes = {"sax": (32, 32, 4), "lax_2c": (32, 32)}
    for v in views:
        t[f"fwd/image/{v}"] = images[v]
    t["fwd/cls_logits"] = classification_forward(model, images, sizes, torch.bfloat16).detach().float()
    t["fwd/reg_preds"] = regression_forward(model, images, sizes, torch.bfloat16).detach().float()
    whole = {"sax": images["sax"][:, :, :32, :32].contiguous(), "lax_2c": images["lax_2c"]}
    t["fwd/cls_logits_whole"] = classification_forward(model, whole, sizes, torch.bfloat16).detach().float()
    save_file({k: v.detach().clone().contiguous() for k, v in t.items()}, str(OUT / "convvit_heads.safetensors"))
    print("wrote convvit_heads.safetensors", {k: tuple(v.shape) for k, v in t.items() if not k.startswith(("cls/grad", "reg/grad"))})
    print("cls loss", float(t["cls/loss"]), "reg", t["reg/metrics"].tolist(), "fwd", t["fwd/cls_logits"].tolist(), t["fwd/reg_preds"].tolist())


if __name__ == "__main__":
    main()
