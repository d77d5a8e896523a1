"""Golden vectors for the ConvViT classifier path, generated from the upstream reference
(runs ONLY where /root/reference exists):  python oracle/make_golden_convvit.py  ->  tests/golden/convvit_*.{safetensors,json}

Fixtures are data (weights, inputs, injected stem masks, expected logits / features / gradients, parameter-group tables,
checkpoint-loading outcomes); no reference source is copied.
"""

from __future__ import annotations

import json
import sys
import tempfile
from pathlib import Path

import torch
from safetensors.torch import save_file

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import ref_shim  # noqa: E402

ref_shim.install()

from cinema.convvit import ConvViT, load_pretrain_weights, param_groups_lr_decay  # noqa: E402
from cinema.mae.mae import CineMA  # noqa: E402

OUT = HERE.parent / "tests" / "golden"


def convvit_kwargs() -> dict:
    views = ["sax", "lax_2c"]
    return dict(image_size_dict={"sax": (32, 32, 4), "lax_2c": (32, 32)}, in_chans_dict=dict.fromkeys(views, 1), n_frames=2, out_chans=3,
                enc_patch_size_dict={"sax": (2, 2, 1), "lax_2c": (2, 2)}, enc_scale_factor_dict={"sax": (2, 2, 1), "lax_2c": (2, 2)},
                enc_conv_chans=[8, 16], enc_conv_n_blocks=1, enc_embed_dim=32, enc_depth=2, enc_n_heads=2)


def fixed_masks(batch: int, n_patches: int, ratio: float, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    noise = torch.rand(batch, n_patches, generator=g)
    rank = torch.argsort(torch.argsort(noise, dim=1), dim=1)
    return rank >= int(n_patches * (1 - ratio))


def main() -> None:
    torch.set_num_threads(8)
    kw = convvit_kwargs()
    torch.manual_seed(0)
    model = ConvViT(**kw)
    model.eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    t = {f"param/{k}": v for k, v in sd.items()}
    g = torch.Generator().manual_seed(3)
    images = {"sax": torch.rand(2, 2, 32, 32, 4, generator=g), "lax_2c": torch.rand(2, 2, 32, 32, generator=g)}
    grids = {"sax": (4, 4, 4), "lax_2c": (4, 4)}
    masks = {v: fixed_masks(2, int(torch.tensor(grids[v]).prod()), 0.5, 11 + i) for i, v in enumerate(images)}
    for v in images:
        t[f"image/{v}"] = images[v]
        t[f"mask/{v}"] = masks[v].to(torch.uint8)
    for tag, md in (("nomask", None), ("mask", masks)):
        feats = model.feature_forward(images, md)
        for k, f in feats.items():
            t[f"feature_{tag}/{k}"] = f.detach()
        for reduce in ("patch", "all", "cls"):
            t[f"logits_{tag}/{reduce}"] = model(images, md, reduce=reduce).detach()
    # gradients of reduce="all" with the stem masks, loss = sum(logits * coef)
    coef = torch.tensor([[0.5, -1.0, 2.0], [1.5, 0.25, -0.75]])
    t["grad/coef"] = coef
    model.zero_grad()
    (model(images, masks, reduce="all") * coef).sum().backward()
    for name in ("pred_head_dict.sax.weight", "pred_head_dict.cls.bias", "encoder.blocks.1.mlp.fc1.weight", "enc_fusion_dict.lax_2c.down_convs.0.weight",
                 "enc_down_dict.sax.conv_blocks.0.patch_embed.conv.weight", "enc_down_dict.sax.conv_blocks.1.conv.0.dw_conv.weight",
                 "enc_down_dict.lax_2c.linear.bias", "encoder.cls_token"):
        t[f"grad/{name}"] = dict(model.named_parameters())[name].grad.detach().clone()
    save_file({k: v.detach().clone().contiguous() for k, v in t.items()}, str(OUT / "convvit_mini.safetensors"))

    meta: dict = {"kwargs": {k: (v if not isinstance(v, dict) else {a: list(b) if isinstance(b, tuple) else b for a, b in v.items()}) for k, v in kw.items()}}
    # layer-decay parameter groups
    groups = param_groups_lr_decay(model, no_weight_decay_list=[], weight_decay=0.05, layer_decay=0.75)
    names = {id(p): n for n, p in model.named_parameters()}
    meta["param_groups"] = [{"lr_scale": gr["lr_scale"], "weight_decay": gr["weight_decay"], "params": [names[id(p)] for p in gr["params"]]} for gr in groups]
    # MAE checkpoint -> ConvViT (single-frame MAE stem filters are tiled over the 2 frames); which tensors arrive, which stay at init
    torch.manual_seed(1)
    mae = CineMA(image_size_dict=kw["image_size_dict"], in_chans_dict=kw["in_chans_dict"], enc_patch_size_dict=kw["enc_patch_size_dict"],
                 enc_scale_factor_dict=kw["enc_scale_factor_dict"], enc_conv_chans=kw["enc_conv_chans"], enc_conv_n_blocks=kw["enc_conv_n_blocks"],
                 enc_embed_dim=32, enc_depth=2, enc_n_heads=2, dec_embed_dim=16, dec_depth=1, dec_n_heads=2)
    mae_sd = {k: v.detach().clone().contiguous() for k, v in mae.state_dict().items()}
    save_file(mae_sd, str(OUT / "convvit_mae_ckpt.safetensors"))
    with tempfile.TemporaryDirectory() as d:
        ck = Path(d) / "mae.safetensors"
        save_file(mae_sd, str(ck))
        torch.manual_seed(2)
        fresh = ConvViT(**kw)
        before = {k: v.detach().clone() for k, v in fresh.state_dict().items()}
        loaded = load_pretrain_weights(fresh, views=["sax", "lax_2c"], ckpt_path=ck, freeze=True)
        after = loaded.state_dict()
        changed = sorted(k for k in after if not torch.equal(after[k], before[k]))
        meta["load_pretrain"] = {"changed": changed, "frozen": sorted(n for n, p in loaded.named_parameters() if not p.requires_grad)}
        w = after["enc_down_dict.sax.conv_blocks.0.patch_embed.conv.weight"]
        meta["load_pretrain"]["stem_tiled_equal"] = bool(torch.equal(w[:, 0], w[:, 1]) and torch.equal(w[:, :1], mae_sd["enc_down_dict.sax.conv_blocks.0.patch_embed.conv.weight"]))
        try:
            load_pretrain_weights(ConvViT(**kw), views=["sax"], ckpt_path=ck, freeze=False)
            meta["load_pretrain"]["single_view_into_two_view_model"] = "ok"
        except ValueError as e:
            meta["load_pretrain"]["single_view_into_two_view_model"] = "ValueError: " + str(e)[:60]
    (OUT / "convvit_meta.json").write_text(json.dumps(meta, indent=1))
    print("wrote convvit_mini.safetensors", sum(v.numel() for v in t.values()) * 4 / 1e6, "MB")


if __name__ == "__main__":
    main()
