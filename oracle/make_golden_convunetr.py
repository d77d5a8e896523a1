"""Golden vectors for the ConvUNetR segmentation path, generated from the upstream reference
(runs ONLY where /root/reference exists):  python oracle/make_golden_convunetr.py  ->  tests/golden/convunetr_*.{safetensors,json}
Fixtures are data (weights, inputs, expected logits and gradients, a ConvResBlock / UpsampleDecoder layer KAT); no reference source is copied.
"""

from __future__ import annotations

import json
import sys
from pathlib import Path

import torch
from safetensors.torch import save_file

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import ref_shim  # noqa: E402

ref_shim.install()

from cinema.conv import ConvResBlock  # noqa: E402
from cinema.segmentation.convunetr import ConvUNetR, UpsampleDecoder, check_conv_unetr_enc_dec_compatiblity  # noqa: E402

OUT = HERE.parent / "tests" / "golden"


def kwargs() -> dict:
    return dict(image_size_dict={"sax": (64, 64, 4), "lax_4c": (64, 64)}, in_chans_dict={"sax": 1, "lax_4c": 1}, out_chans=4,
                enc_patch_size_dict={"sax": (4, 4, 1), "lax_4c": (4, 4)}, enc_scale_factor_dict={"sax": (2, 2, 1), "lax_4c": (2, 2)},
                enc_conv_chans=[8, 16], enc_conv_n_blocks=1, enc_embed_dim=32, enc_depth=2, enc_n_heads=2, dec_chans=(8, 8, 16, 32, 64),
                dec_patch_size_dict={"sax": (2, 2, 1), "lax_4c": (2, 2)}, dec_scale_factor_dict={"sax": (2, 2, 1), "lax_4c": (2, 2)})


def main() -> None:
    torch.set_num_threads(8)
    kw = kwargs()
    torch.manual_seed(0)
    model = ConvUNetR(**kw)
    model.eval()
    t = {f"param/{k}": v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    images = {"sax": torch.rand(2, 1, 64, 64, 4, generator=g), "lax_4c": torch.rand(2, 1, 64, 64, generator=g)}
    for v, im in images.items():
        t[f"image/{v}"] = im
    logits = model(images)
    coefs = {v: torch.randn(logits[v].shape, generator=g) for v in images}
    for v in images:
        t[f"logits/{v}"] = logits[v].detach()
        t[f"coef/{v}"] = coefs[v]
    sum((logits[v] * coefs[v]).sum() for v in images).backward()
    named = dict(model.named_parameters())
    for name in ("pred_head_dict.sax.weight", "decoder_dict.sax.blocks.0.up.weight", "decoder_dict.sax.blocks.4.conv.1.conv2.weight",
                 "decoder_dict.lax_4c.blocks.2.conv.0.norm1.weight", "dec_image_conv_block_dict.sax.conv1.weight", "dec_image_conv_block_dict.sax.norm1.bias",
                 "dec_image_conv_block_dict.sax.shortcut.weight", "dec_conv_blocks_dict.sax.0.conv1.weight", "dec_conv_blocks_dict.lax_4c.3.shortcut.bias",
                 "dec_down_blocks_dict.sax.0.weight", "encoder.blocks.0.attn.kv.weight", "enc_down_dict.sax.conv_blocks.0.conv.0.dw_conv.weight",
                 "enc_down_dict.lax_4c.linear.weight"):
        t[f"grad/{name}"] = named[name].grad.detach().clone()
    # layer KATs: a ConvResBlock with a channel change (3-D, 5 -> 16 channels: exercises the non-multiple-of-8 im2col path) and an UpsampleDecoder
    torch.manual_seed(1)
    blk = ConvResBlock(n_dims=3, in_chans=5, out_chans=16, norm="layer")
    x = torch.randn(2, 5, 6, 7, 3, generator=g)
    for k, v in blk.state_dict().items():
        t[f"resblock/param/{k}"] = v.detach().clone()
    t["resblock/x"], t["resblock/y"] = x, blk(x).detach()
    dec = UpsampleDecoder(n_dims=2, chans=(8, 16), patch_size=(2, 2), scale_factor=(2, 2), norm="layer")
    emb = [torch.randn(1, 8, 8, 8, generator=g), None, torch.randn(1, 16, 2, 2, generator=g)]
    for k, v in dec.state_dict().items():
        t[f"updec/param/{k}"] = v.detach().clone()
    t["updec/e0"], t["updec/e2"] = emb[0], emb[2]
    t["updec/y"] = dec(list(emb)).detach()
    save_file({k: v.detach().clone().contiguous() for k, v in t.items()}, str(OUT / "convunetr_mini.safetensors"))
    meta = {"kwargs": {k: ({a: list(b) for a, b in v.items()} if isinstance(v, dict) and isinstance(next(iter(v.values())), tuple) else
                           (list(v) if isinstance(v, tuple) else v)) for k, v in kw.items()},
            "n_layers_wo_skip": model.n_layers_wo_skip, "n_downsample_layers": len(model.dec_down_blocks_dict["sax"]),
            "compat": {"acdc": list(check_conv_unetr_enc_dec_compatiblity((4, 4, 1), (2, 2, 1), 2, 5, (2, 2, 1), (2, 2, 1)))},
            "state_dict": {k: list(v.shape) for k, v in model.state_dict().items()}}
    (OUT / "convunetr_meta.json").write_text(json.dumps(meta, indent=0))
    print("wrote convunetr_mini.safetensors", sum(v.numel() for v in t.values()) * 4 / 1e6, "MB; wo_skip", model.n_layers_wo_skip)


def fingerprint(sd: dict) -> dict:
    return {k: {"shape": list(v.shape), "sum": float(v.double().sum()), "abs": float(v.double().abs().sum()), "head": [float(x) for x in v.flatten()[:4]]}
            for k, v in sd.items()}


def kwargs_mid() -> dict:
    """MFMA-sized ConvUNetR: the ACDC decoder widths (32 .. 512 channels, cinema/segmentation/acdc/config.yaml:56-66) on a 64 x 64 x 4 volume, ViT
    E = 256 / head_dim 64, 64- / 128-channel stem."""
    return dict(image_size_dict={"sax": (64, 64, 4)}, in_chans_dict={"sax": 1}, out_chans=4, enc_patch_size_dict={"sax": (4, 4, 1)},
                enc_scale_factor_dict={"sax": (2, 2, 1)}, enc_conv_chans=[64, 128], enc_conv_n_blocks=1, enc_embed_dim=256, enc_depth=2, enc_n_heads=4,
                dec_chans=(32, 64, 128, 256, 512), dec_patch_size_dict={"sax": (2, 2, 1)}, dec_scale_factor_dict={"sax": (2, 2, 1)})


def gen_mid() -> None:
    """Reference logits and gradients WITHOUT the weights (seeded construction is bit-identical in the build; fingerprint stored and checked)."""
    torch.set_num_threads(8)
    kw = kwargs_mid()
    torch.manual_seed(0)
    model = ConvUNetR(**kw)
    model.eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(15)
    image = torch.rand(2, 1, 64, 64, 4, generator=g)
    logits = model({"sax": image})["sax"]
    coef = torch.randn(logits.shape, generator=g)
    (logits * coef).sum().backward()
    t = {"image/sax": image, "logits/sax": logits.detach(), "coef/sax": coef}
    named = dict(model.named_parameters())
    t["grad_sq_norm"] = sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None).float().reshape(1)
    for name in ("pred_head_dict.sax.weight", "decoder_dict.sax.blocks.0.up.weight", "decoder_dict.sax.blocks.3.conv.1.conv2.weight",
                 "decoder_dict.sax.blocks.1.conv.0.conv1.weight", "dec_image_conv_block_dict.sax.conv1.weight", "dec_image_conv_block_dict.sax.shortcut.weight",
                 "dec_conv_blocks_dict.sax.0.conv1.weight", "dec_conv_blocks_dict.sax.2.conv2.weight", "dec_conv_blocks_dict.sax.1.norm2.bias",
                 "dec_down_blocks_dict.sax.0.weight", "encoder.blocks.0.attn.kv.weight", "encoder.blocks.1.mlp.fc1.weight",
                 "enc_down_dict.sax.conv_blocks.0.conv.0.dw_conv.weight", "enc_down_dict.sax.linear.weight"):
        gk = named[name].grad.detach().reshape(named[name].shape[0], -1) if named[name].dim() > 1 else named[name].grad.detach()
        t[f"grad/{name}"] = gk[::4].clone() if gk.numel() >= 65536 else gk.clone()
    save_file({k: v.detach().clone().contiguous() for k, v in t.items()}, str(OUT / "convunetr_mid.safetensors"))
    (OUT / "convunetr_mid_meta.json").write_text(json.dumps({"seed_init": 0, "grad_row_stride_large": 4, "large_numel": 65536, "params": fingerprint(sd)}, indent=0))
    print("wrote convunetr_mid.safetensors", sum(v.numel() * v.element_size() for v in t.values()) / 1e6, "MB")


if __name__ == "__main__":
    main()
    gen_mid()
