"""The CPU oracle (oracle/cinema_oracle.py) against golden vectors captured from the upstream reference.

Pins the oracle.  CPU only.
"""

from __future__ import annotations

import json
import math

import pytest
import torch

import cinema_oracle as O
from conftest import GOLDEN, load_golden

torch.set_num_threads(8)


def split(t: dict, prefix: str) -> dict:
    return {k[len(prefix):]: v for k, v in t.items() if k.startswith(prefix)}


def tiny_cfg() -> O.MAEConfig:
    return O.mae_config("tiny", sax_size=(128, 128, 8), views=("sax",))


def mini_cfg(**kw) -> O.MAEConfig:  # noqa: ANN003
    views = ["sax", "lax_2c", "lax_3c", "lax_4c"]
    return O.MAEConfig(
        image_size_dict={v: (32, 32, 4) if v == "sax" else (32, 32) for v in views},
        in_chans_dict=dict.fromkeys(views, 1),
        enc_patch_size_dict={v: (4, 4, 1) if v == "sax" else (4, 4) for v in views},
        enc_scale_factor_dict={v: (2, 2, 1) if v == "sax" else (2, 2) for v in views},
        enc_conv_chans=[16, 32], enc_conv_n_blocks=1, enc_embed_dim=64, enc_depth=2, enc_n_heads=4,
        dec_embed_dim=32, dec_depth=2, dec_n_heads=4, **kw)


def run_model(cfg: O.MAEConfig, g: dict, params: dict, tol: float = 2e-5) -> None:
    p = {k: v.clone().requires_grad_(not k.endswith("pos_embed")) for k, v in params.items()}
    images = split(g, "image/")
    masks = {k: v.bool() for k, v in split(g, "mask/").items()}
    loss, preds, metrics = O.mae_forward(p, cfg, images, masks)
    assert torch.allclose(loss, g["loss"][0], rtol=tol, atol=tol)
    for v, t in split(g, "pred/").items():
        assert preds[v].shape == t.shape
        assert torch.allclose(preds[v], t, rtol=1e-4, atol=tol), v
    for k, t in split(g, "metric/").items():
        assert torch.allclose(metrics[k], t[0], rtol=1e-4, atol=tol), k
    loss.backward()
    for k, t in split(g, "grad/").items():
        assert p[k].grad is not None, k
        assert torch.allclose(p[k].grad, t, rtol=2e-3, atol=2e-6), (k, (p[k].grad - t).abs().max())
    sq = sum((v.grad.double() ** 2).sum() for v in p.values() if v.grad is not None)
    assert abs(float(sq) - float(g["grad_sq_norm"][0])) <= 1e-3 * float(g["grad_sq_norm"][0])


def test_tiny_cfg1_forward_backward() -> None:
    g = load_golden("tiny_sax.safetensors")
    run_model(tiny_cfg(), g, split(g, "param/"))


def test_mini_4view_forward_backward_and_features() -> None:
    g = load_golden("mini_4view.safetensors")
    params = split(g, "param/")
    run_model(mini_cfg(), g, params)
    feats = O.feature_forward(params, mini_cfg(), split(g, "image/"))
    for k, t in split(g, "feature/").items():
        assert torch.allclose(feats[k], t, rtol=1e-4, atol=2e-5), k


@pytest.mark.parametrize(("name", "kw"), [("mini_4view_selfattn", {"cross_attn": False}), ("mini_4view_normtarget", {"norm_target": True})])
def test_mini_variants(name: str, kw: dict) -> None:
    params = split(load_golden("mini_4view.safetensors"), "param/")
    run_model(mini_cfg(**kw), load_golden(f"{name}.safetensors"), params)


def test_param_shapes_match_reference_manifests() -> None:
    man = json.loads((GOLDEN / "state_dict_manifests.json").read_text())
    for name, size, sax, lax in [("base_4view_192", "base", (192, 192, 16), (192, 192)), ("large_4view_256", "large", (256, 256, 24), (256, 256)),
                                 ("base_4view_refdefault", "base", (192, 192, 16), (256, 256))]:
        cfg = O.mae_config(size, sax_size=sax, lax_size=lax)
        shapes = {k: list(s) for k, (s, _) in O.param_shapes(cfg).items()}
        assert shapes == man[name]["keys"], name
        assert list(shapes) == list(man[name]["keys"])  # same ordering as the reference state_dict


def test_weight_decay_groups_match_timm_split() -> None:
    man = json.loads((GOLDEN / "state_dict_manifests.json").read_text())["base_4view_192"]
    cfg = O.mae_config("base", sax_size=(192, 192, 16), lax_size=(192, 192))
    p = {k: torch.empty(s, device="meta") for k, (s, _) in O.param_shapes(cfg).items()}
    groups = O.weight_decay_groups(p, 0.05)
    assert sorted(groups[0]["params"]) == sorted(man["no_decay"])
    assert "encoder.cls_token" in groups[1]["params"]
    assert sum(p[k].numel() for g in groups for k in g["params"]) == man["n_trainable"]


def test_three_step_trajectory() -> None:
    g = load_golden("tiny_sax_trajectory.safetensors")
    params = split(load_golden("tiny_sax.safetensors"), "param/")
    tr = O.Trainer(params, tiny_cfg(), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05, clip_grad=5.0)
    gen = torch.Generator().manual_seed(7)
    for i in range(3):
        lr = O.lr_at(i / 4, 1, 5, 1e-3, 1e-6)
        assert lr == pytest.approx(float(g[f"step{i}/lr"][0]), rel=1e-12, abs=1e-15)
        tr.set_lr(lr)
        image = torch.rand(2, 1, 128, 128, 8, generator=gen)
        assert torch.equal(image.flatten()[:64], g[f"step{i}/image_head"])
        loss, norm, _, _ = tr.step({"sax": image}, {"sax": g[f"step{i}/mask"].bool()})
        assert torch.allclose(loss, g[f"step{i}/loss"][0], rtol=1e-4, atol=1e-5)
        assert torch.allclose(norm, g[f"step{i}/grad_norm"][0], rtol=1e-3, atol=1e-5)
        for k, t in split(g, f"step{i}/param/").items():
            assert torch.allclose(tr.p[k].detach(), t, rtol=1e-3, atol=2e-6), (i, k)


def test_lr_schedule_table() -> None:
    for row in json.loads((GOLDEN / "lr_schedule.json").read_text()):
        assert O.lr_at(*row["args"]) == pytest.approx(row["lr"], rel=1e-12, abs=1e-18)


def test_layer_kats() -> None:
    g = load_golden("layers.safetensors")
    for hd, heads in [(8, 2), (32, 2), (64, 1)]:
        p = {f"a.{k}": v for k, v in split(g, f"attn{hd}/param/").items()}
        q, k = g[f"attn{hd}/q"], g[f"attn{hd}/k"]
        assert torch.allclose(O.attention(q, q, p, "a", heads), g[f"attn{hd}/self"], rtol=1e-4, atol=1e-5)
        assert torch.allclose(O.attention(q, k, p, "a", heads), g[f"attn{hd}/cross"], rtol=1e-4, atol=1e-5)
    # reference quirk: rotary indexed by head cancels in q.k^T -> identical to no rotary
    p = {f"a.{k}": v for k, v in split(g, "rotary/param/").items()}
    assert torch.allclose(g["rotary/out"], g["rotary/out_plain"], atol=1e-6)
    assert torch.allclose(O.attention(g["rotary/x"], g["rotary/x"], p, "a", 4), g["rotary/out"], rtol=1e-4, atol=1e-5)
    assert g["rotary/cos"].shape == (4, 4)  # rows = n_heads, cols = head_dim / 2
    for nd in (2, 3):
        p = {f"b.{k}": v for k, v in split(g, f"mcb{nd}d/param/").items()}
        x, vis = g[f"mcb{nd}d/x"], g[f"mcb{nd}d/vis"].bool()
        assert torch.allclose(O.masked_conv_block(x, vis, p, "b"), g[f"mcb{nd}d/out_masked"], rtol=1e-4, atol=1e-5)
        assert torch.allclose(O.masked_conv_block(x, None, p, "b"), g[f"mcb{nd}d/out"], rtol=1e-4, atol=1e-5)
    for nd, size in [(2, (32, 32)), (3, (32, 32, 4))]:
        cfg = O.MAEConfig(image_size_dict={"v": size}, in_chans_dict={"v": 1}, enc_patch_size_dict={"v": (4, 4, 1)[:nd]},
                          enc_scale_factor_dict={"v": (2, 2, 1)[:nd]}, enc_conv_chans=[8, 16], enc_conv_n_blocks=1, enc_embed_dim=24,
                          enc_depth=0, enc_n_heads=1, dec_embed_dim=8, dec_depth=0, dec_n_heads=1)
        p = {f"d.{k}": v for k, v in split(g, f"down{nd}d/param/").items()}
        p.update({f"f.{k}": v for k, v in split(g, f"fuse{nd}d/param/").items()})
        m = g[f"down{nd}d/mask"].bool()
        skips, tok = O.downsample_encoder(g[f"down{nd}d/image"], m, p, "d", cfg, "v")
        assert torch.allclose(tok, g[f"down{nd}d/tokens"], rtol=1e-4, atol=1e-5)
        for i, s in enumerate(skips):
            assert torch.allclose(s, g[f"down{nd}d/skip{i}"], rtol=1e-4, atol=1e-5)
        _, tok2 = O.downsample_encoder(g[f"down{nd}d/image_other"], None, p, "d", cfg, "v")
        assert torch.allclose(tok2, g[f"down{nd}d/tokens_other"], rtol=1e-4, atol=1e-5)  # pos-embed interpolation
        kept = tok[~m].reshape(2, -1, 24)
        assert torch.allclose(O.multi_scale_fusion(skips, kept, m, p, "f", 1e-5), g[f"fuse{nd}d/out_masked"], rtol=1e-4, atol=1e-5)
        assert torch.allclose(O.multi_scale_fusion(skips, tok, None, p, "f", 1e-5), g[f"fuse{nd}d/out_full"], rtol=1e-4, atol=1e-5)
    for name, dim, grid in [("sax768", 768, (12, 12, 16)), ("sax512", 512, (12, 12, 16)), ("lax768", 768, (12, 12)), ("odd", 20, (2, 3, 4)),
                            ("odd2d", 10, (3, 2))]:
        pe = O.sincos_pos_embed(dim, grid)
        pe = pe[:, ::37] if pe.shape[1] > 64 else pe
        assert torch.allclose(pe, g[f"pos_embed/{name}"], atol=1e-6), name
    for i in range(4):
        out = O.upsample_mask(g[f"upsample_mask/{i}/in"].bool(), tuple(g[f"upsample_mask/{i}/scale"].tolist()))
        assert torch.equal(out, g[f"upsample_mask/{i}/out"].bool())
    for nt in (0, 1):
        loss, metrics = O.mse_loss(g["mse/target"], g["mse/pred"], g["mse/mask"].bool(), bool(nt))
        assert torch.allclose(loss, g[f"mse/{nt}/loss"][0], rtol=1e-5)
        for k, v in metrics.items():
            assert torch.allclose(v, g[f"mse/{nt}/{k}"][0], rtol=1e-5), k
    assert torch.equal(O.patchify(g["patchify/image3d"], (2, 3, 1)), g["patchify/out3d"])
    assert torch.equal(O.patchify(g["patchify/image2d"], (2, 2)), g["patchify/out2d"])
    assert torch.equal(O.unpatchify(g["patchify/out3d"], (2, 3, 1), (2, 2, 2)), g["patchify/image3d"])
    assert torch.equal(O.unpatchify(g["patchify/out2d"], (2, 2), (2, 3)), g["patchify/image2d"])


def test_reference_value_tables() -> None:
    """Value tests restated from the reference's own suite (convvit_test.py:21-50, mae_test.py:15-32, vit errors)."""
    m = torch.tensor([[[True, False], [False, True]]])
    out = O.upsample_mask(m, (2, 2))
    expect = torch.tensor([[[1, 1, 0, 0], [1, 1, 0, 0], [0, 0, 1, 1], [0, 0, 1, 1]]]).bool()
    assert torch.equal(out, expect)
    for n, r in [(16, 0.75), (10, 0.5), (7, 0.3), (512, 0.75)]:
        mask = O.random_patch_mask(3, n, r)
        assert mask.shape == (3, n)
        assert int((~mask).sum()) == int(n * (1 - r)) * 3
    assert not O.random_patch_mask(2, 5, 0.0).any()
    with pytest.raises(ValueError):
        O.random_patch_mask(2, 5, -0.1)
    with pytest.raises(ValueError):
        O.patchify(torch.zeros(1, 1, 5, 4), (2, 2))
    with pytest.raises(ValueError):
        O.unpatchify(torch.zeros(1, 4, 8), (2, 2), (2, 3))


# ------------------------------------------------------------------------------------------------ ConvViT
def convvit_cfg() -> O.MAEConfig:
    meta = json.loads((GOLDEN / "convvit_meta.json").read_text())["kwargs"]
    return O.MAEConfig(image_size_dict=meta["image_size_dict"], in_chans_dict=meta["in_chans_dict"], enc_patch_size_dict=meta["enc_patch_size_dict"],
                       enc_scale_factor_dict=meta["enc_scale_factor_dict"], enc_conv_chans=meta["enc_conv_chans"],
                       enc_conv_n_blocks=meta["enc_conv_n_blocks"], enc_embed_dim=meta["enc_embed_dim"], enc_depth=meta["enc_depth"],
                       enc_n_heads=meta["enc_n_heads"], dec_embed_dim=16, dec_depth=1, dec_n_heads=2)


def test_convvit_logits_features_and_gradients() -> None:
    """Oracle restatement of ``ConvViT.feature_forward`` / ``forward`` against the reference (all reduce modes, with and without stem masks)."""
    g = load_golden("convvit_mini.safetensors")
    cfg = convvit_cfg()
    params = {k: v.clone().requires_grad_(not k.endswith("pos_embed")) for k, v in split(g, "param/").items()}
    images = split(g, "image/")
    masks = {k: v.bool() for k, v in split(g, "mask/").items()}
    for tag, md in (("nomask", None), ("mask", masks)):
        feats = O.convvit_features(params, cfg, images, md)
        for k, t in split(g, f"feature_{tag}/").items():
            assert torch.allclose(feats[k], t, rtol=1e-4, atol=2e-5), (tag, k)
        for reduce, t in split(g, f"logits_{tag}/").items():
            out = O.convvit_forward(params, cfg, images, md, reduce=reduce)
            assert out.shape == t.shape and torch.allclose(out, t, rtol=1e-4, atol=2e-5), (tag, reduce)
    (O.convvit_forward(params, cfg, images, masks, reduce="all") * g["grad/coef"]).sum().backward()
    for k, t in split(g, "grad/").items():
        if k == "coef":
            continue
        assert torch.allclose(params[k].grad, t, rtol=1e-3, atol=1e-6), k
    with pytest.raises(NotImplementedError):
        O.convvit_forward(params, cfg, images, None, reduce="none")


def test_convvit_head_losses_and_patch_averaged_forward() -> None:
    """Oracle restatement of the fine-tuning heads against the reference's ``classification_loss`` / ``regression_loss`` (values, reported
    metrics, gradients through the ConvViT oracle) and ``classification_forward`` / ``regression_forward`` (one over-sized view, half-overlapping patches)."""
    g, h = load_golden("convvit_mini.safetensors"), load_golden("convvit_heads.safetensors")
    cfg = convvit_cfg()
    params = {k: v.clone().requires_grad_(not k.endswith("pos_embed")) for k, v in split(g, "param/").items()}
    images = split(g, "image/")
    logits = O.convvit_forward(params, cfg, images)
    assert torch.allclose(logits, h["cls/logits"], rtol=1e-4, atol=2e-5)
    loss = O.classification_loss_value(logits, h["cls/label"], 0.1)
    assert abs(float(loss) - float(h["cls/loss"])) <= 1e-5 and abs(float(loss) - float(h["cls/metrics"][0])) <= 1e-5
    loss.backward()
    for k, t in split(h, "cls/grad/").items():
        assert torch.allclose(params[k].grad, t, rtol=1e-3, atol=1e-6), k
    for p_ in params.values():
        p_.grad = None
    preds = O.convvit_forward(params, cfg, images)
    vals = O.regression_loss_values(preds.detach(), h["reg/label"])
    for i, k in enumerate(("mse_loss", "mae_loss", "max_label", "min_label", "max_pred", "min_pred", "loss")):
        assert abs(vals[k] - float(h["reg/metrics"][i])) <= 1e-5, k
    ((preds - h["reg/label"]) ** 2).mean().backward()
    for k, t in split(h, "reg/grad/").items():
        assert torch.allclose(params[k].grad, t, rtol=1e-3, atol=1e-6), k
    fwd_images, sizes = split(h, "fwd/image/"), {"sax": (32, 32, 4), "lax_2c": (32, 32)}
    with torch.no_grad():
        fwd = lambda d: O.convvit_forward(params, cfg, d)  # noqa: E731
        assert torch.allclose(O.patch_average_forward(fwd, fwd_images, sizes, "classification"), h["fwd/cls_logits"], rtol=1e-4, atol=2e-5)
        assert torch.allclose(O.patch_average_forward(fwd, fwd_images, sizes, "regression"), h["fwd/reg_preds"], rtol=1e-4, atol=2e-5)
        whole = {"sax": fwd_images["sax"][:, :, :32, :32].contiguous(), "lax_2c": fwd_images["lax_2c"]}
        assert torch.allclose(O.patch_average_forward(fwd, whole, sizes, "classification"), h["fwd/cls_logits_whole"], rtol=1e-4, atol=2e-5)


# ------------------------------------------------------------------------------------------------ ConvUNetR
def _unetr_setup():  # noqa: ANN202
    meta = json.loads((GOLDEN / "convunetr_meta.json").read_text())
    kw = meta["kwargs"]
    cfg = O.MAEConfig(image_size_dict=kw["image_size_dict"], in_chans_dict=kw["in_chans_dict"], enc_patch_size_dict=kw["enc_patch_size_dict"],
                      enc_scale_factor_dict=kw["enc_scale_factor_dict"], enc_conv_chans=kw["enc_conv_chans"], enc_conv_n_blocks=kw["enc_conv_n_blocks"],
                      enc_embed_dim=kw["enc_embed_dim"], enc_depth=kw["enc_depth"], enc_n_heads=kw["enc_n_heads"], dec_embed_dim=16, dec_depth=1,
                      dec_n_heads=2)
    return meta, cfg


def test_convunetr_logits_gradients_and_layer_kats() -> None:
    """Oracle restatement of ``ConvUNetR.forward`` / ``ConvResBlock`` / ``UpsampleDecoder`` against the reference."""
    g = load_golden("convunetr_mini.safetensors")
    meta, cfg = _unetr_setup()
    params = {k: v.clone().requires_grad_(not k.endswith("pos_embed")) for k, v in split(g, "param/").items() if not k.startswith(("resblock", "updec"))}
    images = split(g, "image/")
    out = O.convunetr_forward(params, cfg, tuple(meta["kwargs"]["dec_chans"]), meta["n_layers_wo_skip"], meta["n_downsample_layers"], images)
    for v, t in split(g, "logits/").items():
        assert out[v].shape == t.shape and torch.allclose(out[v], t, rtol=1e-4, atol=5e-5), v
    sum((out[v] * g[f"coef/{v}"]).sum() for v in images).backward()
    for k, t in split(g, "grad/").items():
        assert torch.allclose(params[k].grad, t, rtol=2e-3, atol=1e-5 * float(t.abs().max()) + 1e-7), k
    rp = split(g, "resblock/param/")
    assert torch.allclose(O.conv_res_block(g["resblock/x"], {f"b.{k}": v for k, v in rp.items()}, "b"), g["resblock/y"], rtol=1e-4, atol=2e-5)
    up = {f"d.{k}": v for k, v in split(g, "updec/param/").items()}
    y = O.upsample_decoder([g["updec/e0"], None, g["updec/e2"]], up, "d", 2)
    assert torch.allclose(y, g["updec/y"], rtol=1e-4, atol=2e-5)


def test_segmentation_loss_known_answer() -> None:
    """Hand-computed value (monai is absent: the Dice restatement is unpinned against the reference, see the oracle docstring).
    2 classes, 1 sample, 4 voxels, zero logits -> p = 0.5 everywhere; labels [1, 1, 0, -1]:
      CE over the 3 labelled voxels = ln 2;  foreground class: I = 1.0, P = 2.0, G = 2  ->  dice = 1 - (2 + 1e-5) / (4 + 1e-5)."""
    logits = torch.zeros(1, 2, 4, requires_grad=True)
    labels = torch.tensor([[[1, 1, 0, -1]]])
    loss, m = O.segmentation_loss_one_view(logits, labels)
    ce, dice = math.log(2.0), 1.0 - (2.0 + 1e-5) / (4.0 + 1e-5)
    assert float(m["cross_entropy"]) == pytest.approx(ce, rel=1e-6) and float(m["mean_dice_loss"]) == pytest.approx(dice, rel=1e-6)
    assert float(loss) == pytest.approx(ce + dice, rel=1e-6)
    loss.backward()
    assert torch.isfinite(logits.grad).all() and float(logits.grad[0, :, 3].abs().sum()) > 0  # the ignored voxel still feeds the Dice sums


# ------------------------------------------------------------------------------------------------ evaluation path
# known answers of the reference's own test (cinema/transform_test.py:13-96): (patch_size, image_size, patch_overlap) -> start indices
_GRID_KATS = [
    ((3, 5), (3, 5), (0, 0), [[0, 0]]),
    ((3, 5), (3, 7), (0, 0), [[0, 0], [0, 2]]),
    ((4, 5, 6), (8, 10, 6), (2, 1, 4), [[0, 0, 0], [0, 4, 0], [0, 5, 0], [2, 0, 0], [2, 4, 0], [2, 5, 0], [4, 0, 0], [4, 4, 0], [4, 5, 0]]),
    ((128, 128, 128), (192, 128, 128), (64, 0, 0), [[0, 0, 0], [64, 0, 0]]),
]


def test_patch_grid_and_aggregation_vs_reference() -> None:
    """Oracle ``patch_grid`` and the product's ``get_patch_grid`` / ``patch_grid_sample`` / ``aggregate_patches`` (host helpers) against the
    reference's known answers and generated vectors (oracle/make_golden_seg_eval.py)."""
    import numpy as np

    from cinema_amd.transform import aggregate_patches, crop_start, get_patch_grid, patch_grid_sample

    for patch, size, ov, want in _GRID_KATS:
        assert O.patch_grid(size, patch, ov).tolist() == want
        assert get_patch_grid(size, patch, ov).tolist() == want
    g = load_golden("seg_eval.safetensors")
    for i in range(4):
        a = g[f"grid/{i}/args"].tolist()
        n = len(a) // 3
        size, patch, ov = tuple(a[:n]), tuple(a[n:2 * n]), tuple(a[2 * n:])
        assert O.patch_grid(size, patch, ov).tolist() == g[f"grid/{i}/starts"].tolist()
        assert get_patch_grid(size, patch, ov).tolist() == g[f"grid/{i}/starts"].tolist()
    with pytest.raises(ValueError, match="should be <= image size"):
        get_patch_grid((4, 4), (5, 4), (0, 0))
    starts = get_patch_grid((8, 10, 6), (4, 5, 6), (2, 1, 4))
    assert torch.equal(aggregate_patches(g["agg/patches"], starts, (8, 10, 6)), g["agg/out"])
    img = aggregate_patches(g["agg/sampled"], starts, (8, 10, 6))  # windows of one image average back to the image (transform_test.py:180-182)
    assert torch.equal(patch_grid_sample(img, starts, (4, 5, 6)), g["agg/sampled"])
    assert patch_grid_sample(img[0], starts, (4, 5, 6)).shape == (9, 4, 5, 6)
    assert crop_start(np.zeros((2, 3, 4)), (1, 2, 3)).shape == (1, 2, 3)
    with pytest.raises(ValueError, match="same length"):
        crop_start(np.zeros((2, 3)), (1, 2, 3))


def test_sliding_window_forward_vs_reference() -> None:
    """Oracle ``sliding_window_logits`` around the oracle ConvUNetR against the reference's ``segmentation_forward`` on its own model (12 windows)."""
    g = load_golden("seg_eval.safetensors")
    meta, cfg = _unetr_setup()
    params = {k: v for k, v in split(load_golden("convunetr_mini.safetensors"), "param/").items() if not k.startswith(("resblock", "updec"))}

    def fwd(images: dict) -> dict:
        with torch.no_grad():
            return O.convunetr_forward(params, cfg, tuple(meta["kwargs"]["dec_chans"]), meta["n_layers_wo_skip"], meta["n_downsample_layers"], images)

    images = split(g, "fwd/image/")
    out = O.sliding_window_logits(fwd, images, {"sax": (64, 64, 4), "lax_4c": (64, 64)})
    for v, t in split(g, "fwd/logits/").items():
        assert out[v].shape == t.shape and torch.allclose(out[v], t, rtol=1e-4, atol=1e-4), (v, float((out[v] - t).abs().max()))
    whole = {"sax": images["sax"][:, :, :64, :64, :4].contiguous(), "lax_4c": images["lax_4c"]}
    out2 = O.sliding_window_logits(fwd, whole, {"sax": (64, 64, 4), "lax_4c": (64, 64)})
    assert torch.allclose(out2["sax"], g["fwd/whole_logits/sax"], rtol=1e-4, atol=1e-4)
    with pytest.raises(ValueError, match="smaller than patch size"):
        O.sliding_window_logits(fwd, whole, {"sax": (64, 64, 8), "lax_4c": (64, 64)})


# cinema/metric_test.py:60-75 (default threshold 0, offset 1): logits (1, 3, 2, 2) -> stability per class
_STABILITY_KATS = [
    ([[[[0.8, 2.3], [-1.0, 1.1]], [[-1.8, -2.0], [1.5, -1.3]], [[1.0, -0.3], [-0.5, 0.2]]]], [0.5, 1.0, 0.25]),
    ([[[[0.8, 1.3], [-1.0, 1.1]], [[-0.8, -3.0], [1.5, -1.3]], [[2.0, -1.3], [-0.5, 0.2]]]], [0.5, 1.0, 0.25]),
]


def test_segmentation_metrics_oracle_known_answers_and_independent_dice() -> None:
    """stability_score against the reference's own known answers (pins the compute_iou restatement); Dice / IoU / volumes against an
    independent float64 derivation by explicit voxel counting (monai absent: compute_dice itself stays unpinned against the reference);
    the Dice LOSS restatement against a second derivation in float64 with explicit loops."""
    for logits, want in _STABILITY_KATS:
        got = O.stability_score(torch.tensor(logits))
        assert torch.allclose(torch.nan_to_num(got), torch.tensor([want]), rtol=1e-5, atol=1e-5)
    torch.manual_seed(0)
    logits = torch.randn(2, 4, 6, 5, 3)
    labels = torch.randint(0, 3, (2, 1, 6, 5, 3))  # class 3 never occurs: NaN scores (ignore_empty)
    m = O.segmentation_metrics(logits, labels, (1.5, 1.5, 10.0))
    pred = logits.argmax(1)
    for b in range(2):
        for k in (1, 2, 3):
            p, t = (pred[b] == k), (labels[b, 0] == k)
            inter, ps, ts = float((p & t).sum()), float(p.sum()), float(t.sum())
            if ts == 0:
                assert math.isnan(float(m[f"class_{k}_dice_score"][b])) and math.isnan(float(m[f"class_{k}_iou_score"][b]))
            else:
                assert float(m[f"class_{k}_dice_score"][b]) == pytest.approx(2 * inter / (ps + ts), rel=1e-6)
                assert float(m[f"class_{k}_iou_score"][b]) == pytest.approx(inter / (ps + ts - inter), rel=1e-6)
            assert float(m[f"class_{k}_true_volume"][b]) == pytest.approx(ts * 22.5 / 1000.0, rel=1e-6)
            assert float(m[f"class_{k}_pred_volume"][b]) == pytest.approx(ps * 22.5 / 1000.0, rel=1e-6)
    assert math.isnan(float(m["mean_dice_score"][0]))
    # Hausdorff distance (monai absent: unpinned against the reference): known answers of the restatement and an independent brute-force derivation
    z = torch.zeros(1, 12, 12, 4, dtype=torch.long)
    a, b_ = z.clone(), z.clone()
    a[0, 2:6, 2:6, 1:3] = 1
    b_[0, 5:9, 2:6, 1:3] = 1                      # the same box moved by 3 voxels along x: every surface point is 3 voxels from the other surface or closer
    hd = O.hausdorff_distance_95(a, b_, 2, (1.5, 1.0, 10.0))
    assert float(hd[0, 0]) == pytest.approx(4.5, rel=1e-6) and math.isnan(float(hd[0, 1]))   # 3 voxels x 1.5 mm; class 2 absent on both sides
    assert math.isinf(float(O.hausdorff_distance_95(a, z, 1, (1.0, 1.0, 1.0))[0, 0]))           # one side empty
    torch.manual_seed(3)
    pl, tl = torch.randint(0, 3, (2, 9, 8, 5)), torch.randint(0, 3, (2, 9, 8, 5))
    sp = (1.25, 0.8, 6.0)
    hd = O.hausdorff_distance_95(pl, tl, 2, sp)
    for b in range(2):
        for k in (1, 2):
            def surface(lab: torch.Tensor) -> torch.Tensor:  # face-neighbour definition, explicit loops
                m_, pts = lab == k, []
                for i in range(9):
                    for j in range(8):
                        for l_ in range(5):
                            if not m_[i, j, l_]:
                                continue
                            nb = [(i - 1, j, l_), (i + 1, j, l_), (i, j - 1, l_), (i, j + 1, l_), (i, j, l_ - 1), (i, j, l_ + 1)]
                            if not all(0 <= x < 9 and 0 <= y < 8 and 0 <= w < 5 and bool(m_[x, y, w]) for x, y, w in nb):
                                pts.append((i * sp[0], j * sp[1], l_ * sp[2]))
                return torch.tensor(pts, dtype=torch.float64)
            sa, sb = surface(pl[b]), surface(tl[b])
            d = torch.cdist(sa, sb)
            want = max(float(torch.quantile(d.min(1).values, 0.95)), float(torch.quantile(d.min(0).values, 0.95)))
            assert float(hd[b, k - 1]) == pytest.approx(want, rel=1e-5)
    # Dice loss, second derivation: float64, explicit loops over samples / classes / voxels
    lg = torch.randn(2, 3, 4, 5, dtype=torch.float64)
    lb = torch.randint(-1, 3, (2, 1, 4, 5))
    _, mm = O.segmentation_loss_one_view(lg.float(), lb)
    prob = torch.softmax(lg, dim=1)
    terms = []
    for b in range(2):
        for c in (1, 2):
            inter = den = 0.0
            for i in range(4):
                for j in range(5):
                    t = 1.0 if max(int(lb[b, 0, i, j]), 0) == c else 0.0
                    inter += float(prob[b, c, i, j]) * t
                    den += float(prob[b, c, i, j]) + t
            terms.append(1.0 - (2.0 * inter + 1e-5) / (den + 1e-5))
    assert float(mm["mean_dice_loss"]) == pytest.approx(sum(terms) / len(terms), rel=1e-5)


def test_oracle_at_mfma_sized_channels_vs_reference_golden() -> None:
    """The oracle pinned at the channel counts the MFMA kernels use (E = 256 / head_dim 64, decoder 128 / head_dim 32, 64- / 128-channel stem, batch 3):
    tests/golden/midsize_2view.safetensors (oracle/make_golden.py::gen_midsize, written by the upstream reference).  Weights = the seeded
    construction of the build, whose fingerprint must equal the reference's."""
    from cinema_amd import CineMA

    g = load_golden("midsize_2view.safetensors")
    meta = json.loads((GOLDEN / "midsize_2view_meta.json").read_text())
    views = ["sax", "lax_2c"]
    kw = dict(image_size_dict={"sax": (64, 64, 8), "lax_2c": (64, 64)}, in_chans_dict=dict.fromkeys(views, 1),
              enc_patch_size_dict={"sax": (4, 4, 1), "lax_2c": (4, 4)}, enc_scale_factor_dict={"sax": (2, 2, 1), "lax_2c": (2, 2)},
              enc_conv_chans=[64, 128], enc_conv_n_blocks=1, enc_embed_dim=256, enc_depth=2, enc_n_heads=4, dec_embed_dim=128, dec_depth=2, dec_n_heads=4)
    torch.manual_seed(meta["seed_init"])
    sd = {k: v.detach().clone() for k, v in CineMA(**kw).state_dict().items()}
    for k, v in sd.items():
        f = meta["params"][k]
        assert list(v.shape) == f["shape"] and [float(x) for x in v.flatten()[:4]] == f["head"], k
        assert abs(float(v.double().sum()) - f["sum"]) <= 1e-6 * max(1.0, f["abs"]), k
    p = {k: v.clone().requires_grad_(not k.endswith("pos_embed")) for k, v in sd.items()}
    loss, preds, metrics = O.mae_forward(p, O.MAEConfig(**kw), split(g, "image/"), {k: v.bool() for k, v in split(g, "mask/").items()})
    assert torch.allclose(loss, g["loss"][0], rtol=2e-5, atol=2e-5)
    for v, t in split(g, "pred/").items():
        assert torch.allclose(preds[v], t, rtol=1e-4, atol=5e-5), v
    loss.backward()
    sq = sum(float(q.grad.double().pow(2).sum()) for q in p.values() if q.grad is not None)
    assert abs(sq - float(g["grad_sq_norm"][0])) <= 1e-4 * float(g["grad_sq_norm"][0])
    for k, t in split(g, "grad/").items():
        gk = p[k].grad.reshape(p[k].shape[0], -1) if p[k].dim() > 1 else p[k].grad
        gk = gk[::meta["grad_row_stride_large"]] if gk.numel() >= meta["large_numel"] else gk
        assert torch.allclose(gk, t, rtol=2e-3, atol=2e-5 * float(t.abs().max()) + 1e-8), (k, float((gk - t).abs().max()), float(t.abs().max()))


def test_convunetr_oracle_at_mfma_sized_channels_vs_reference_golden() -> None:
    """``O.convunetr_forward`` with the ACDC decoder widths (32 .. 512 channels) against tests/golden/convunetr_mid.safetensors (upstream reference,
    oracle/make_golden_convunetr.py::gen_mid); weights from the seeded construction, fingerprint checked."""
    from cinema_amd.segmentation.convunetr import ConvUNetR

    g = load_golden("convunetr_mid.safetensors")
    meta = json.loads((GOLDEN / "convunetr_mid_meta.json").read_text())
    kw = dict(image_size_dict={"sax": (64, 64, 4)}, in_chans_dict={"sax": 1}, out_chans=4, enc_patch_size_dict={"sax": (4, 4, 1)},
              enc_scale_factor_dict={"sax": (2, 2, 1)}, enc_conv_chans=[64, 128], enc_conv_n_blocks=1, enc_embed_dim=256, enc_depth=2, enc_n_heads=4,
              dec_chans=(32, 64, 128, 256, 512), dec_patch_size_dict={"sax": (2, 2, 1)}, dec_scale_factor_dict={"sax": (2, 2, 1)})
    torch.manual_seed(meta["seed_init"])
    sd = {k: v.detach().clone() for k, v in ConvUNetR(**kw).state_dict().items()}
    for k, v in sd.items():
        f = meta["params"][k]
        assert list(v.shape) == f["shape"] and [float(x) for x in v.flatten()[:4]] == f["head"], k
    cfg = O.MAEConfig(image_size_dict=kw["image_size_dict"], in_chans_dict=kw["in_chans_dict"], enc_patch_size_dict=kw["enc_patch_size_dict"],
                      enc_scale_factor_dict=kw["enc_scale_factor_dict"], enc_conv_chans=kw["enc_conv_chans"], enc_conv_n_blocks=kw["enc_conv_n_blocks"],
                      enc_embed_dim=kw["enc_embed_dim"], enc_depth=kw["enc_depth"], enc_n_heads=kw["enc_n_heads"], dec_embed_dim=16, dec_depth=1, dec_n_heads=2)
    p = {k: v.clone().requires_grad_(not k.endswith("pos_embed")) for k, v in sd.items()}
    logits = O.convunetr_forward(p, cfg, kw["dec_chans"], 1, 1, {"sax": g["image/sax"]})["sax"]
    assert torch.allclose(logits, g["logits/sax"], rtol=1e-4, atol=1e-4 * float(g["logits/sax"].abs().max()))
    (logits * g["coef/sax"]).sum().backward()
    for k, t in split(g, "grad/").items():
        gk = p[k].grad.reshape(p[k].shape[0], -1) if p[k].dim() > 1 else p[k].grad
        gk = gk[::meta["grad_row_stride_large"]] if gk.numel() >= meta["large_numel"] else gk
        assert torch.allclose(gk, t, rtol=5e-3, atol=5e-5 * float(t.abs().max()) + 1e-8), (k, float((gk - t).abs().max()), float(t.abs().max()))


def test_monai_restatements_agree_with_the_independent_second_statement() -> None:
    """Rows a26 / f3: monai 1.5.2 is absent, the reference holds no value for ``DiceLoss`` or for ``Zoom`` / ``ScaleIntensity`` / ``SpatialPad``.  The oracle's
    restatement (torch tensor ops) and ``oracle/second_opinion.py`` (float64 loops over samples / classes / voxels and explicit index maps, written from monai's
    published algorithm, sharing no code with the oracle) must agree on the committed vectors ``tests/golden/second_opinion.safetensors``: five random
    segmentation cases + an absent class + ignored voxels + an all-background volume + a 2-D case; trilinear / bicubic zoom in and out on odd extents, the
    identity zoom and a constant image.  Both sides are recomputed here and compared with the stored values."""
    import numpy as np

    import second_opinion as S

    g = load_golden("second_opinion.safetensors")
    seg = sorted({k.split("/")[1] for k in g if k.startswith("seg/")})
    assert len(seg) >= 9
    for name in seg:
        logits, labels, want = g[f"seg/{name}/logits"], g[f"seg/{name}/labels"].long(), g[f"seg/{name}/values"]
        _, m = O.segmentation_loss_one_view(logits.double(), labels)
        s = S.segmentation_loss(logits.numpy().astype(np.float64), labels.numpy())
        for i, k in enumerate(("cross_entropy", "mean_dice_loss", "loss")):
            assert abs(float(m[k]) - float(want[i])) <= 1e-9 and abs(s[k] - float(want[i])) <= 1e-9, (name, k)
        _, m32 = O.segmentation_loss_one_view(logits, labels)  # the fp32 form the GPU tests compare against
        assert abs(float(m32["mean_dice_loss"]) - float(want[1])) <= 2e-6
    # the absent class: its term is 1 - smooth / (sum of its probabilities + smooth), i.e. the loss does not collapse to "perfect" for an empty class
    lg, lb = g["seg/absent_class/logits"], g["seg/absent_class/labels"].long()
    assert int(lb.max()) == 2 and lg.shape[1] == 4
    tf = sorted({k.split("/")[1] for k in g if k.startswith("tf/")})
    assert len(tf) >= 8
    for name in tf:
        x, want, args = g[f"tf/{name}/x"], g[f"tf/{name}/y"], g[f"tf/{name}/args"]
        zoom, cubic, padded = float(args[0]), bool(args[1]), tuple(int(v) for v in args[2:])
        got = O.input_transform(x.double(), zoom, padded, cubic)
        ref = S.input_transform(x.numpy().astype(np.float64), zoom, padded, cubic)
        assert tuple(got.shape) == tuple(want.shape) == ref.shape
        assert float((got - want).abs().max()) <= 1e-9 and float(np.abs(ref - want.numpy()).max()) <= 1e-12, name
    const = torch.full((5, 4, 3), -1.5)
    assert float(O.input_transform(const, 1.0, (6, 4, 4), False).abs().max()) == 0.0 and float(np.abs(S.input_transform(const.numpy(), 1.0, (6, 4, 4), False)).max()) == 0.0
