"""Generate golden vectors from the upstream reference (runs ONLY where /root/reference exists).

Usage:  python oracle/make_golden.py            # writes tests/golden/*.safetensors / *.json

The fixtures are data (inputs, injected masks, weights, expected outputs); no reference source is
copied.  The reference is imported in place through ``oracle/ref_shim.py``.
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

from cinema.conv import MaskedConvBlock  # noqa: E402
from cinema.convvit import DownsampleEncoder, MultiScaleFusion, upsample_mask  # noqa: E402
from cinema.mae import mae as ref_mae  # noqa: E402
from cinema.mae.mae import CineMA, mse_loss  # noqa: E402
from cinema.optim import GradScaler, adjust_learning_rate  # noqa: E402
from cinema.vit import Attention, get_pos_embed, get_vit_config, patchify  # noqa: E402
from timm.optim import param_groups_weight_decay  # noqa: E402

OUT = HERE.parent / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)


def fixed_masks(batch: int, n_patches: int, ratio: float, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    noise = torch.rand(batch, n_patches, generator=g)
    rank = torch.argsort(torch.argsort(noise, dim=1), dim=1)
    return rank >= int(n_patches * (1 - ratio))


class InjectMasks:
    """Patch the reference's mask sampler so identical masks can be replayed elsewhere."""

    def __init__(self, masks: list) -> None:
        self.masks = list(masks)
        self.i = 0

    def __enter__(self):  # noqa: ANN204
        self.orig = ref_mae.get_batch_random_patch_mask

        def fake(batch_size, n_patches, mask_ratio, device):  # noqa: ANN001, ANN202, ARG001
            m = self.masks[self.i % len(self.masks)]
            self.i += 1
            assert m.shape == (batch_size, n_patches)
            return m.clone()

        ref_mae.get_batch_random_patch_mask = fake
        return self

    def __exit__(self, *a):  # noqa: ANN002, ANN204
        ref_mae.get_batch_random_patch_mask = self.orig


def model_kwargs(name: str) -> dict:
    if name == "tiny_sax":  # BASELINE.json configs[0]
        return dict(image_size_dict={"sax": (128, 128, 8)}, in_chans_dict={"sax": 1}, enc_patch_size_dict={"sax": (4, 4, 1)},
                    enc_scale_factor_dict={"sax": (2, 2, 1)}, enc_conv_chans=[64, 128], enc_conv_n_blocks=2, **get_vit_config("tiny"))
    if name == "mini_4view":
        views = ["sax", "lax_2c", "lax_3c", "lax_4c"]
        return dict(
            image_size_dict={v: (32, 32, 4) if v == "sax" else (32, 32) for v in views},
            in_chans_dict=dict.fromkeys(views, 1),
            enc_patch_size_dict={v: (4, 4, 1) if v == "sax" else (4, 4) for v in views},
            enc_scale_factor_dict={v: (2, 2, 1) if v == "sax" else (2, 2) for v in views},
            enc_conv_chans=[16, 32], enc_conv_n_blocks=1, enc_embed_dim=64, enc_depth=2, enc_n_heads=4,
            dec_embed_dim=32, dec_depth=2, dec_n_heads=4)
    if name == "midsize_2view":  # MFMA-sized channels: E = 256 / head_dim 64, decoder 128 / head_dim 32, 64- / 128-channel stem
        views = ["sax", "lax_2c"]
        return dict(
            image_size_dict={"sax": (64, 64, 8), "lax_2c": (64, 64)}, in_chans_dict=dict.fromkeys(views, 1),
            enc_patch_size_dict={"sax": (4, 4, 1), "lax_2c": (4, 4)}, enc_scale_factor_dict={"sax": (2, 2, 1), "lax_2c": (2, 2)},
            enc_conv_chans=[64, 128], enc_conv_n_blocks=1, enc_embed_dim=256, enc_depth=2, enc_n_heads=4, dec_embed_dim=128, dec_depth=2, dec_n_heads=4)
    raise KeyError(name)


def gen_midsize() -> None:
    """Reference outputs at MFMA-sized channel counts WITHOUT the weights: the seeded construction is bit-identical between the reference and the
    build (fingerprint stored and checked), so the fixture holds the seed, inputs, masks, loss, predictions, metrics, the squared gradient norm and the
    gradients of a dozen tensors (every 4th row of the large matrices)."""
    name, batch, seed_init, seed_data = "midsize_2view", 3, 0, 21
    kw = model_kwargs(name)
    torch.manual_seed(seed_init)
    model = CineMA(**kw)
    model.train()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(seed_data)
    images = {v: torch.rand(batch, 1, *kw["image_size_dict"][v], generator=g) for v in kw["image_size_dict"]}
    masks = {v: fixed_masks(batch, model.enc_down_dict[v].patch_embed.n_patches, 0.75, 300 + i) for i, v in enumerate(images)}
    with InjectMasks([masks[v] for v in images]):
        loss, pred, _, metrics = model(images, 0.75)
    loss.backward()
    t = {f"image/{v}": x for v, x in images.items()}
    t.update({f"mask/{v}": x.to(torch.uint8) for v, x in masks.items()})
    t.update({f"pred/{v}": x.detach() for v, x in pred.items()})
    t.update({f"metric/{k}": x.detach().reshape(1) for k, x in metrics.items()})
    t["loss"] = loss.detach().reshape(1)
    t["grad_sq_norm"] = sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None).float().reshape(1)
    keys = ["enc_down_dict.sax.conv_blocks.0.conv.0.dw_conv.weight", "enc_down_dict.sax.conv_blocks.1.conv.0.mlp.fc1.weight",
            "enc_down_dict.sax.conv_blocks.1.patch_embed.conv.weight", "enc_down_dict.sax.patch_embed.proj.weight", "enc_down_dict.lax_2c.linear.weight",
            "encoder.blocks.0.attn.q.weight", "encoder.blocks.0.attn.kv.weight", "encoder.blocks.1.attn.proj.weight", "encoder.blocks.1.mlp.fc1.weight",
            "encoder.blocks.0.mlp.fc2.weight", "encoder.blocks.0.norm1.weight", "encoder.blocks.1.mlp.fc1.bias", "encoder.norm.bias", "encoder.cls_token",
            "enc_fusion_dict.sax.down_convs.0.weight", "dec_linear.weight", "decoder.blocks.0.attn.q.weight", "decoder.blocks.1.attn.kv.weight",
            "decoder.blocks.1.mlp.fc2.weight", "decoder.blocks.0.norm2.bias", "dec_embed_dict.sax.mask_token", "pred_head_dict.lax_2c.weight",
            "pred_head_dict.sax.bias"]
    named = dict(model.named_parameters())
    for k in keys:
        gk = named[k].grad.detach().reshape(named[k].shape[0], -1) if named[k].dim() > 1 else named[k].grad.detach()
        t[f"grad/{k}"] = gk[::4].clone() if gk.numel() >= 65536 else gk.clone()
    save_file({k: v.detach().clone().contiguous() for k, v in t.items()}, str(OUT / f"{name}.safetensors"))
    (OUT / f"{name}_meta.json").write_text(json.dumps({"seed_init": seed_init, "seed_data": seed_data, "batch": batch, "grad_row_stride_large": 4,
                                                         "large_numel": 65536, "params": fingerprint(sd)}, indent=0))
    print(name, "loss", float(loss), "tensors", len(t), "MB", sum(v.numel() * v.element_size() for v in t.values()) / 1e6)


GRAD_KEYS = {
    "tiny_sax": ["encoder.blocks.0.attn.kv.weight", "enc_down_dict.sax.conv_blocks.0.conv.0.dw_conv.weight",
                 "dec_embed_dict.sax.mask_token", "pred_head_dict.sax.weight", "encoder.cls_token",
                 "enc_down_dict.sax.conv_blocks.0.patch_embed.conv.weight", "enc_fusion_dict.sax.down_convs.0.weight",
                 "decoder.blocks.0.norm1.weight", "enc_down_dict.sax.conv_blocks.1.conv.1.mlp.fc1.bias"],
    "mini_4view": None,  # all
}


def fingerprint(sd: dict) -> dict:
    return {k: {"shape": list(v.shape), "sum": float(v.double().sum()), "abs": float(v.double().abs().sum()),
                "head": [float(x) for x in v.flatten()[:4]]} for k, v in sd.items()}


def gen_model(name: str, batch: int, seed_init: int, seed_data: int, cross_attn: bool = True, norm_target: bool = False) -> None:
    kw = model_kwargs(name)
    torch.manual_seed(seed_init)
    model = CineMA(**kw, cross_attn=cross_attn, norm_target=norm_target)
    model.train()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(seed_data)
    images = {v: torch.rand(batch, 1, *kw["image_size_dict"][v], generator=g) for v in kw["image_size_dict"]}
    masks = {v: fixed_masks(batch, model.enc_down_dict[v].patch_embed.n_patches, 0.75, 100 + i) for i, v in enumerate(images)}
    with InjectMasks([masks[v] for v in images]):
        loss, pred, mask_out, metrics = model(images, 0.75)
    for v in images:
        assert torch.equal(mask_out[v], masks[v])
    loss.backward()
    tensors = {}
    suffix = ("" if cross_attn else "_selfattn") + ("_normtarget" if norm_target else "")
    if suffix == "":
        tensors.update({f"param/{k}": v for k, v in sd.items()})
    tensors.update({f"image/{v}": t for v, t in images.items()})
    tensors.update({f"mask/{v}": t.to(torch.uint8) for v, t in masks.items()})
    tensors.update({f"pred/{v}": t.detach() for v, t in pred.items()})
    tensors.update({f"metric/{k}": t.detach().reshape(1) for k, t in metrics.items()})
    tensors["loss"] = loss.detach().reshape(1)
    keys = GRAD_KEYS[name] or [k for k, p in model.named_parameters() if p.grad is not None]
    named = dict(model.named_parameters())
    for k in keys:
        tensors[f"grad/{k}"] = named[k].grad.detach().clone()
    tensors["grad_sq_norm"] = sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None).float().reshape(1)
    if suffix == "":
        model.eval()
        with torch.no_grad():
            feats = model.feature_forward(images)
        tensors.update({f"feature/{k}": t for k, t in feats.items()})
    save_file({k: v.detach().clone().contiguous() for k, v in tensors.items()}, str(OUT / f"{name}{suffix}.safetensors"))
    if suffix == "":
        (OUT / f"{name}_init_fingerprint.json").write_text(json.dumps({"seed": seed_init, "params": fingerprint(sd)}, indent=0))
    print(name + suffix, "loss", float(loss), "n_tensors", len(tensors))


def gen_trajectory() -> None:
    """G7: 3 optimisation steps of the reference harness semantics on cfg 1 (CPU fp32)."""
    kw = model_kwargs("tiny_sax")
    torch.manual_seed(0)
    model = CineMA(**kw)
    model.train()
    opt = torch.optim.AdamW(param_groups_weight_decay(model, 0.05), lr=1e-3, betas=(0.9, 0.95))
    scaler = GradScaler()
    g = torch.Generator().manual_seed(7)
    track = ["encoder.blocks.0.attn.q.weight", "enc_down_dict.sax.conv_blocks.0.conv.0.dw_conv.weight",
             "dec_embed_dict.sax.mask_token", "pred_head_dict.sax.bias"]
    named = dict(model.named_parameters())
    tensors = {}  # initial weights == tiny_sax.safetensors param/* (same seed-0 construction); images from Generator(7)
    n_iter = 4  # pretend epoch length so that the lr moves
    for i in range(3):
        lr = adjust_learning_rate(opt, step=i / n_iter + 0, warmup_steps=1, max_n_steps=5, lr=1e-3, min_lr=1e-6)
        images = {"sax": torch.rand(2, 1, 128, 128, 8, generator=g)}
        masks = {"sax": fixed_masks(2, 512, 0.75, 200 + i)}
        with InjectMasks([masks["sax"]]):
            loss, _, _, _ = model(images, 0.75)
        norm = scaler(loss=loss, optimizer=opt, clip_grad=5.0, parameters=model.parameters(), update_grad=True)
        opt.zero_grad()
        tensors[f"step{i}/image_head"] = images["sax"].flatten()[:64].clone()  # RNG-drift check; full image = 3 draws of Generator(7)
        tensors[f"step{i}/mask"] = masks["sax"].to(torch.uint8)
        tensors[f"step{i}/loss"] = loss.detach().reshape(1)
        tensors[f"step{i}/grad_norm"] = norm.detach().reshape(1)
        tensors[f"step{i}/lr"] = torch.tensor([lr], dtype=torch.float64)
        for k in track:
            tensors[f"step{i}/param/{k}"] = named[k].detach().clone()
        print("traj step", i, float(loss), float(norm), lr)
    save_file({k: v.detach().clone().contiguous() for k, v in tensors.items()}, str(OUT / "tiny_sax_trajectory.safetensors"))
    table = []
    for step, warm, mx, lr, mn in [(0.0, 10, 800, 1e-3, 1e-6), (3.5, 10, 800, 1e-3, 1e-6), (10.0, 10, 800, 1e-3, 1e-6),
                                   (123.25, 10, 800, 1e-3, 1e-6), (799.9, 10, 800, 1e-3, 1e-6), (2.0, 0, 4, 5e-4, 0.0)]:
        o = torch.optim.SGD([{"params": [torch.zeros(1, requires_grad=True)]},
                             {"params": [torch.zeros(1, requires_grad=True)], "lr_scale": 0.5}], lr=0.1)
        val = adjust_learning_rate(o, step, warm, mx, lr, mn)
        table.append({"args": [step, warm, mx, lr, mn], "lr": val, "group_lrs": [pg["lr"] for pg in o.param_groups]})
    (OUT / "lr_schedule.json").write_text(json.dumps(table, indent=1))


def gen_layers() -> None:
    """G3: per-layer known-answer vectors."""
    t: dict = {}
    torch.manual_seed(11)
    for hd, heads, tq, tk in [(8, 2, 9, 9), (32, 2, 10, 7), (64, 1, 5, 12)]:
        dim = hd * heads
        attn = Attention(dim, n_heads=heads, qkv_bias=True)
        for k, v in attn.state_dict().items():
            t[f"attn{hd}/param/{k}"] = v.clone()
        q, k = torch.randn(2, tq, dim), torch.randn(2, tk, dim)
        t[f"attn{hd}/q"], t[f"attn{hd}/k"] = q, k
        t[f"attn{hd}/self"] = attn(q).detach()
        t[f"attn{hd}/cross"] = attn(q, k).detach()
    # rotary quirk: rotation indexed by head cancels in q.k^T
    attn = Attention(32, n_heads=4, qkv_bias=True, rotary=True)
    attn_plain = Attention(32, n_heads=4, qkv_bias=True, rotary=False)
    attn_plain.load_state_dict(attn.state_dict())
    x = torch.randn(2, 6, 32)
    t["rotary/x"], t["rotary/out"], t["rotary/out_plain"] = x, attn(x).detach(), attn_plain(x).detach()
    t["rotary/cos"] = attn.rotary.cos.detach().clone()
    for k, v in attn.state_dict().items():
        t[f"rotary/param/{k}"] = v.clone()
    for nd, size in [(2, (6, 8)), (3, (6, 4, 5))]:
        blk = MaskedConvBlock(n_dims=nd, in_chans=8)
        for k, v in blk.state_dict().items():
            t[f"mcb{nd}d/param/{k}"] = v.clone()
        x = torch.randn(2, 8, *size)
        vis = torch.rand(2, *size) > 0.4
        t[f"mcb{nd}d/x"], t[f"mcb{nd}d/vis"] = x, vis.to(torch.uint8)
        t[f"mcb{nd}d/out_masked"] = blk(x, vis).detach()
        t[f"mcb{nd}d/out"] = blk(x, None).detach()
    # DownsampleEncoder with a grid different from the built one (pos-embed interpolation) + MultiScaleFusion
    for nd, size, other in [(2, (32, 32), (48, 32)), (3, (32, 32, 4), (32, 48, 6))]:
        ps, sf = (4, 4, 1)[:nd], (2, 2, 1)[:nd]
        enc = DownsampleEncoder(image_size=size, in_chans=1, patch_size=ps, scale_factor=sf, conv_chans=[8, 16], conv_n_blocks=1,
                                embed_dim=24, norm="layer")
        fus = MultiScaleFusion(image_size=size, patch_size=ps, scale_factor=sf, conv_chans=[8, 16], embed_dim=24,
                               norm_layer=torch.nn.LayerNorm, norm_eps=1e-5)
        for k, v in enc.state_dict().items():
            t[f"down{nd}d/param/{k}"] = v.clone()
        for k, v in fus.state_dict().items():
            t[f"fuse{nd}d/param/{k}"] = v.clone()
        img, img2 = torch.rand(2, 1, *size), torch.rand(2, 1, *other)
        n = enc.patch_embed.n_patches
        m = fixed_masks(2, n, 0.5, 5)
        skips, tok = enc(img, m)
        _, tok2 = enc(img2, None)
        kept = tok[~m].reshape(2, -1, 24)
        t[f"down{nd}d/image"], t[f"down{nd}d/image_other"], t[f"down{nd}d/mask"] = img, img2, m.to(torch.uint8)
        t[f"down{nd}d/tokens"], t[f"down{nd}d/tokens_other"] = tok.detach(), tok2.detach()
        for i, s in enumerate(skips):
            t[f"down{nd}d/skip{i}"] = s.detach()
        t[f"fuse{nd}d/out_masked"] = fus(skips, kept, m).detach()
        t[f"fuse{nd}d/out_full"] = fus(skips, tok, None).detach()
    for name, dim, grid in [("sax768", 768, (12, 12, 16)), ("sax512", 512, (12, 12, 16)), ("lax768", 768, (12, 12)), ("odd", 20, (2, 3, 4)),
                            ("odd2d", 10, (3, 2))]:
        pe = get_pos_embed(dim, grid).detach().clone()
        t[f"pos_embed/{name}"] = pe[:, ::37] if pe.shape[1] > 64 else pe  # every 37th row of the big tables keeps the fixture small
    for i, (shape, sf) in enumerate([((2, 3, 2), (2, 2)), ((1, 2, 2, 3), (2, 1, 2)), ((2, 4), (3,)), ((1, 2, 3), (1, 4))]):
        m = torch.rand(*shape) > 0.5
        t[f"upsample_mask/{i}/in"] = m.to(torch.uint8)
        t[f"upsample_mask/{i}/out"] = upsample_mask(m, sf).to(torch.uint8)
        t[f"upsample_mask/{i}/scale"] = torch.tensor(sf)
    target, pred_mask = torch.rand(2, 12, 16), fixed_masks(2, 12, 0.75, 9)
    pred = torch.randn(2, 9, 16)
    for nt in (False, True):
        loss, metrics = mse_loss(target, pred, pred_mask, nt)
        t[f"mse/{int(nt)}/loss"] = loss.reshape(1)
        for k, v in metrics.items():
            t[f"mse/{int(nt)}/{k}"] = v.reshape(1)
    t["mse/target"], t["mse/pred"], t["mse/mask"] = target, pred, pred_mask.to(torch.uint8)
    img = torch.arange(2 * 3 * 4 * 6 * 2, dtype=torch.float32).reshape(2, 3, 4, 6, 2)
    t["patchify/image3d"], t["patchify/out3d"] = img, patchify(img, (2, 3, 1))
    img = torch.arange(2 * 2 * 4 * 6, dtype=torch.float32).reshape(2, 2, 4, 6)
    t["patchify/image2d"], t["patchify/out2d"] = img, patchify(img, (2, 2))
    save_file({k: v.detach().clone().contiguous() for k, v in t.items()}, str(OUT / "layers.safetensors"))
    print("layers", len(t))


def gen_manifests() -> None:
    """G5: state_dict name/shape manifests for the BASELINE configs (no weights)."""
    out = {}
    for name, size, sax, lax in [("base_4view_192", "base", (192, 192, 16), (192, 192)), ("large_4view_256", "large", (256, 256, 24), (256, 256)),
                                 ("base_4view_refdefault", "base", (192, 192, 16), (256, 256))]:
        views = ["sax", "lax_2c", "lax_3c", "lax_4c"]
        with torch.device("meta"):
            m = CineMA(image_size_dict={v: sax if v == "sax" else lax for v in views}, in_chans_dict=dict.fromkeys(views, 1),
                       enc_patch_size_dict={v: (4, 4, 1) if v == "sax" else (4, 4) for v in views},
                       enc_scale_factor_dict={v: (2, 2, 1) if v == "sax" else (2, 2) for v in views},
                       enc_conv_chans=[64, 128], enc_conv_n_blocks=2, **get_vit_config(size))
        out[name] = {"keys": {k: list(v.shape) for k, v in m.state_dict().items()},
                     "n_params": sum(p.numel() for p in m.parameters()),
                     "n_trainable": sum(p.numel() for p in m.parameters() if p.requires_grad),
                     "no_decay": [k for k, p in m.named_parameters() if p.requires_grad and (p.ndim <= 1 or k.endswith(".bias"))]}
        print(name, out[name]["n_params"], out[name]["n_trainable"])
    (OUT / "state_dict_manifests.json").write_text(json.dumps(out))


if __name__ == "__main__":
    torch.set_num_threads(8)
    gen_model("tiny_sax", batch=2, seed_init=0, seed_data=1)
    gen_model("mini_4view", batch=2, seed_init=0, seed_data=2)
    gen_model("mini_4view", batch=2, seed_init=0, seed_data=2, cross_attn=False)
    gen_model("mini_4view", batch=2, seed_init=0, seed_data=2, norm_target=True)
    gen_trajectory()
    gen_layers()
    gen_manifests()
    gen_midsize()
