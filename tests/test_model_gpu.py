"""End-to-end parity of the HIP path (cinema_amd.CineMA on an MI355X) against
 (1) golden vectors captured from the upstream reference (tests/golden, fp32 CPU), and
 (2) the CPU oracle run live on the same seeded inputs.

Tolerances (bf16 MFMA compute, fp32 accumulation / residual stream).  SURVEY.md 8d states loss rel <= 2e-2 and predictions max-abs <= 5e-2;
the bounds below are ~3 x the errors MEASURED on an MI355X (the tests print them with -s; gpurun_out/tests_r02a.log of round 2):

  model (test)                                   loss rel   worst grad rel-L2 (matrix / any)   worst max-abs/max   pred max-abs
  cfg 1 Tiny SAX, reference golden               7.0e-6     0.75 % / 0.75 %                    0.73 %              0.009
  mini 4-view (16/32-channel stem), ref goldens  1.5e-4     3.2 %  / 5.8 % (LN vectors, LAX)   7.5 %               0.018
  midsize 2-view (MFMA-sized) vs live oracle     1.5e-4     1.3 %  / 1.7 %                     3.3 %               0.017
  cfg 2 Base 4-view 192, b=2 vs live oracle      6.9e-4     1.8 %  (10 named: 0.8 %)           2.3 %               0.027

  loss: relative 2e-3 everywhere;  predictions: max-abs 5e-2 (the stated bound);  metrics (fp32 reductions): relative 1e-4;
  gradients: per model class below (the 16-channel mini model's LayerNorm-vector gradients are sums with heavy cancellation: 6 % / 12 %).
"""

from __future__ import annotations

import math

import pytest
import torch

pytestmark = pytest.mark.gpu

import cinema_oracle as O  # noqa: E402
from cinema_amd import CineMA  # noqa: E402
from cinema_amd.vit import get_vit_config  # noqa: E402
from conftest import load_golden  # noqa: E402

DEV = "cuda"
LOSS_RTOL, PRED_ATOL, GRAD_L2, GRAD_MAX = 2e-3, 5e-2, 6e-2, 12e-2   # default gates (the mini goldens carry their own, below)
# mini goldens (16 / 32-channel stems, 8- and 16-wide heads: everything below the MFMA tile sizes).  FIXED bounds since round 6, per class: every weight MATRIX
# within 3 % rel-L2 of the reference's gradient, every VECTOR (LayerNorm / bias: sums over all rows of bf16-rounded gradients) within 5 %; largest element error 6 % of
# the tensor's largest element.  Measured (MI355X, round 6): plain 2.2 % / 2.8 % / 3.9 %, self-attention 2.0 % / 2.5 % / 3.2 %, norm_target (gradients scaled by
# 1 / patch std) 2.8 % / 4.0 % / 4.9 % - the worst tensors are LayerNorm vectors and the 1x1 convolutions of a long-axis stem.
MINI_GATES = {"mini_4view": ((3e-2, 5e-2), 6e-2), "mini_4view_selfattn": ((3e-2, 5e-2), 6e-2), "mini_4view_normtarget": ((3e-2, 5e-2), 6e-2)}
TINY_GRAD_L2, TINY_GRAD_MAX = 2.5e-2, 2.5e-2                          # cfg 1: measured 0.75 % / 0.73 %
MID_GRAD_L2, MID_GRAD_MAX = 5.5e-2, 10e-2                             # MFMA-sized models: measured 1.7 % / 3.3 %
# config-2 real-shape first-step parity (test_base_4view_192_first_step_vs_oracle) = 3 x measured (6.9e-4, 1.2e-3, 5.7e-4, 0.79 %); the worst tensor: a REQUIRED
# 3 % since round 6 (measured 1.46 %, a LayerNorm vector of a long-axis stem; 1.76 % before the attention outputs kept their second half)
CFG2_LOSS_RTOL, CFG2_VIEW_LOSS_RTOL, CFG2_GRAD_NORM_RTOL, CFG2_NAMED_GRAD_L2, CFG2_WORST_GRAD_L2 = 2.1e-3, 3.6e-3, 1.8e-3, 2.4e-2, 3.0e-2


def split(t: dict, prefix: str) -> dict:
    return {k[len(prefix):]: v for k, v in t.items() if k.startswith(prefix)}


def tiny_kwargs() -> dict:
    return dict(image_size_dict={"sax": (128, 128, 8)}, in_chans_dict={"sax": 1}, enc_patch_size_dict={"sax": (4, 4, 1)},
                enc_scale_factor_dict={"sax": (2, 2, 1)}, enc_conv_chans=[64, 128], enc_conv_n_blocks=2, **get_vit_config("tiny"))


def mini_kwargs(**kw) -> dict:  # noqa: ANN003
    views = ["sax", "lax_2c", "lax_3c", "lax_4c"]
    return dict(image_size_dict={v: (32, 32, 4) if v == "sax" else (32, 32) for v in views}, in_chans_dict=dict.fromkeys(views, 1),
                enc_patch_size_dict={v: (4, 4, 1) if v == "sax" else (4, 4) for v in views},
                enc_scale_factor_dict={v: (2, 2, 1) if v == "sax" else (2, 2) for v in views}, enc_conv_chans=[16, 32], enc_conv_n_blocks=1,
                enc_embed_dim=64, enc_depth=2, enc_n_heads=4, dec_embed_dim=32, dec_depth=2, dec_n_heads=4, **kw)


def check_against(model: CineMA, images: dict, masks: dict, ref_loss: torch.Tensor, ref_pred: dict, ref_metrics: dict, ref_grads: dict,
                  grad_l2: float = GRAD_L2, grad_max: float = GRAD_MAX) -> None:
    model.zero_grad(set_to_none=True)
    loss, pred, mask_out, metrics = model({k: v.to(DEV) for k, v in images.items()}, 0.75, enc_mask_dict={k: v.to(DEV) for k, v in masks.items()})
    assert loss.dim() == 0 and torch.isfinite(loss)
    assert abs(float(loss) - float(ref_loss)) <= LOSS_RTOL * abs(float(ref_loss)), (float(loss), float(ref_loss))
    for v, t in ref_pred.items():
        assert pred[v].shape == t.shape, (v, pred[v].shape, t.shape)
        err = (pred[v].float().cpu() - t).abs().max()
        assert err <= PRED_ATOL, (v, float(err))
        assert torch.equal(mask_out[v].cpu(), masks[v])
    for k, t in ref_metrics.items():
        if k.endswith("pred_max"):  # the maximum of the bf16-computed predictions: the prediction bound applies
            assert abs(float(metrics[k]) - float(t)) <= PRED_ATOL, (k, float(metrics[k]), float(t))
            continue
        tol = LOSS_RTOL if (k.endswith("mse_loss") or k == "loss") else 1e-4  # bf16-compute quantities vs fp32 reductions
        assert abs(float(metrics[k]) - float(t)) <= tol * abs(float(t)) + 1e-6, (k, float(metrics[k]), float(t))
    loss.backward()
    named = dict(model.named_parameters())
    worst, worst_max = {}, {}
    for k, t in ref_grads.items():
        g = named[k].grad
        assert g is not None, k
        diff = g.float().cpu() - t
        scale = float(t.abs().max())
        err = float(diff.abs().max())
        l2 = float(diff.norm() / t.norm().clamp_min(1e-12))
        worst[k] = l2
        worst_max[k] = err / max(scale, 1e-30)
        gate = grad_l2 if not isinstance(grad_l2, tuple) else grad_l2[0 if named[k].ndim > 1 else 1]   # (matrices, vectors)
        assert l2 <= gate, (k, l2)
        assert err <= grad_max * scale + 1e-7, (k, err, scale)
    mats = {k: v for k, v in worst.items() if named[k].ndim > 1}
    vecs = {k: v for k, v in worst.items() if named[k].ndim <= 1}
    print("measured: loss rel", abs(float(loss) - float(ref_loss)) / abs(float(ref_loss)), "worst vector rel-L2:", max(vecs.items(), key=lambda kv: kv[1]) if vecs else None,
          "worst max-abs/max:", max(worst_max.items(), key=lambda kv: kv[1]), "worst matrix rel-L2:", max(mats.items(), key=lambda kv: kv[1]) if mats else None,
          "pred max-abs:", max(float((pred[v].float().cpu() - t).abs().max()) for v, t in ref_pred.items()) if ref_pred else None)


def test_tiny_cfg1_vs_reference_golden() -> None:
    g = load_golden("tiny_sax.safetensors")
    model = CineMA(**tiny_kwargs())
    model.load_state_dict(split(g, "param/"))
    model.to(DEV)
    check_against(model, split(g, "image/"), {k: v.bool() for k, v in split(g, "mask/").items()}, g["loss"][0], split(g, "pred/"),
                  {k: v[0] for k, v in split(g, "metric/").items()}, split(g, "grad/"), grad_l2=TINY_GRAD_L2, grad_max=TINY_GRAD_MAX)


@pytest.mark.parametrize(("name", "kw"), [("mini_4view", {}), ("mini_4view_selfattn", {"cross_attn": False}), ("mini_4view_normtarget", {"norm_target": True})])
def test_mini_4view_vs_reference_golden(name: str, kw: dict) -> None:
    g = load_golden(f"{name}.safetensors")
    model = CineMA(**mini_kwargs(**kw))
    model.load_state_dict(split(load_golden("mini_4view.safetensors"), "param/"))
    model.to(DEV)
    check_against(model, split(g, "image/"), {k: v.bool() for k, v in split(g, "mask/").items()}, g["loss"][0], split(g, "pred/"),
                  {k: v[0] for k, v in split(g, "metric/").items()}, split(g, "grad/"), grad_l2=MINI_GATES[name][0], grad_max=MINI_GATES[name][1])


def test_mini_feature_forward_vs_reference_golden() -> None:
    g = load_golden("mini_4view.safetensors")
    model = CineMA(**mini_kwargs())
    model.load_state_dict(split(g, "param/"))
    model.to(DEV).eval()
    with torch.no_grad():
        feats = model.feature_forward({k: v.to(DEV) for k, v in split(g, "image/").items()})
    for k, t in split(g, "feature/").items():
        assert feats[k].shape == t.shape
        assert (feats[k].float().cpu() - t).abs().max() <= 5e-2, k  # LN-normalised O(1) features, bf16 compute


def fingerprint_ok(sd: dict, fp: dict) -> None:
    """The seeded construction reproduces the reference's initial weights (the MFMA-sized fixtures hold no weights, only this fingerprint)."""
    assert set(sd) == set(fp)
    for k, v in sd.items():
        f = fp[k]
        assert list(v.shape) == f["shape"], k
        assert abs(float(v.double().sum()) - f["sum"]) <= 1e-6 * max(1.0, f["abs"]) and abs(float(v.double().abs().sum()) - f["abs"]) <= 1e-6 * max(1.0, f["abs"]), k
        assert [float(x) for x in v.flatten()[:4]] == f["head"], k


def class_bounds(name: str, t: torch.Tensor) -> tuple:
    """Gradient bounds by tensor class: weight matrices 3 % relative L2, vectors (biases, LayerNorm parameters, tokens) 6 %;
    max-abs error <= 2 x the class's relative-L2 bound of the tensor's largest entry."""
    l2 = 3e-2 if (t.dim() > 1 and min(t.shape) > 1 and "token" not in name) else 6e-2
    return l2, 2.0 * l2


def check_grads_by_class(named: dict, ref_grads: dict, meta: dict) -> dict:
    worst = {"matrix": ("", 0.0), "vector": ("", 0.0)}
    for k, t in ref_grads.items():
        g = named[k].grad.float().cpu()
        g = g.reshape(g.shape[0], -1) if g.dim() > 1 else g
        if g.numel() >= meta["large_numel"]:
            g = g[::meta["grad_row_stride_large"]]
        assert g.shape == t.shape, (k, g.shape, t.shape)
        l2 = float((g - t).norm() / t.norm().clamp_min(1e-12))
        mx = float((g - t).abs().max() / t.abs().max().clamp_min(1e-30))
        b_l2, b_mx = class_bounds(k, named[k])
        assert l2 <= b_l2, (k, l2, b_l2)
        assert mx <= b_mx, (k, mx, b_mx)
        cls = "matrix" if b_l2 == 3e-2 else "vector"
        if l2 > worst[cls][1]:
            worst[cls] = (k, l2)
    return worst


def test_midsize_mfma_shapes_vs_reference_golden() -> None:
    """The direct link reference -> HIP path at MFMA-sized channel counts (E = 256 / head_dim 64, decoder 128 / head_dim 32, 64- / 128-channel stem,
    batch 3): tests/golden/midsize_2view.safetensors was written by the upstream reference (oracle/make_golden.py::gen_midsize); weights come from the
    seeded construction (fingerprint checked).  Loss rel <= 2e-3, predictions max-abs <= 5e-2, gradient norm rel <= 1e-2, gradients by class."""
    import json
    from pathlib import Path

    g = load_golden("midsize_2view.safetensors")
    meta = json.loads((Path(__file__).parent / "golden" / "midsize_2view_meta.json").read_text())
    views = ["sax", "lax_2c"]
    kw = dict(image_size_dict={"sax": (64, 64, 8), "lax_2c": (64, 64)}, in_chans_dict=dict.fromkeys(views, 1),
              enc_patch_size_dict={"sax": (4, 4, 1), "lax_2c": (4, 4)}, enc_scale_factor_dict={"sax": (2, 2, 1), "lax_2c": (2, 2)},
              enc_conv_chans=[64, 128], enc_conv_n_blocks=1, enc_embed_dim=256, enc_depth=2, enc_n_heads=4, dec_embed_dim=128, dec_depth=2, dec_n_heads=4)
    torch.manual_seed(meta["seed_init"])
    model = CineMA(**kw)
    fingerprint_ok(model.state_dict(), meta["params"])
    model.to(DEV)
    images, masks = split(g, "image/"), {k: v.bool() for k, v in split(g, "mask/").items()}
    loss, pred, _, metrics = model({k: v.to(DEV) for k, v in images.items()}, 0.75, enc_mask_dict={k: v.to(DEV) for k, v in masks.items()})
    loss.backward()
    rel = abs(float(loss) - float(g["loss"][0])) / float(g["loss"][0])
    assert rel <= LOSS_RTOL, rel
    for v, t in split(g, "pred/").items():
        assert (pred[v].float().cpu() - t).abs().max() <= PRED_ATOL, v
    for k, t in split(g, "metric/").items():
        tol = PRED_ATOL if k.endswith("pred_max") else (LOSS_RTOL if (k.endswith("mse_loss") or k == "loss") else 1e-4)
        assert abs(float(metrics[k]) - float(t[0])) <= tol * abs(float(t[0])) + (PRED_ATOL if k.endswith("pred_max") else 1e-6), k
    named = dict(model.named_parameters())
    gn = math.sqrt(sum(float(p.grad.double().pow(2).sum()) for p in model.parameters() if p.grad is not None))
    gn_ref = math.sqrt(float(g["grad_sq_norm"][0]))
    assert abs(gn - gn_ref) <= 1e-2 * gn_ref, (gn, gn_ref)
    worst = check_grads_by_class(named, split(g, "grad/"), meta)
    print("midsize vs REFERENCE golden: loss rel", rel, "grad norm rel", abs(gn - gn_ref) / gn_ref, "worst by class", worst)


def test_midsize_mfma_path_vs_oracle() -> None:
    """A config whose shapes take the MFMA kernels (E=256/hd=64 encoder, D=128/hd=32 decoder, stem 64/128 channels),
    checked against the CPU oracle on seeded random weights, inputs and masks."""
    views = ["sax", "lax_2c"]
    kw = dict(image_size_dict={"sax": (64, 64, 8), "lax_2c": (64, 64)}, in_chans_dict=dict.fromkeys(views, 1),
              enc_patch_size_dict={"sax": (4, 4, 1), "lax_2c": (4, 4)}, enc_scale_factor_dict={"sax": (2, 2, 1), "lax_2c": (2, 2)},
              enc_conv_chans=[64, 128], enc_conv_n_blocks=1, enc_embed_dim=256, enc_depth=2, enc_n_heads=4, dec_embed_dim=128, dec_depth=2,
              dec_n_heads=4)
    torch.manual_seed(3)
    model = CineMA(**kw)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = O.MAEConfig(**kw)
    gen = torch.Generator().manual_seed(5)
    images = {v: torch.rand(3, 1, *kw["image_size_dict"][v], generator=gen) for v in views}
    masks = {v: O.random_patch_mask(3, math.prod(cfg.grid_size(v)), 0.75, gen) for v in views}
    p = {k: v.clone().requires_grad_(not k.endswith("pos_embed")) for k, v in sd.items()}
    ref_loss, ref_pred, ref_metrics = O.mae_forward(p, cfg, images, masks)
    ref_loss.backward()
    ref_grads = {k: v.grad for k, v in p.items() if v.grad is not None}
    model.to(DEV)
    check_against(model, images, masks, ref_loss.detach(), {k: v.detach() for k, v in ref_pred.items()},
                  {k: v.detach() for k, v in ref_metrics.items()}, ref_grads, grad_l2=MID_GRAD_L2, grad_max=MID_GRAD_MAX)


def test_visible_voxel_stem_equals_dense_stem() -> None:
    """The MAE step evaluates the conv stem on the visible voxels only (cinema_amd/convvit.py); forcing the reference's dense
    evaluation must give the same loss, predictions and parameter gradients up to bf16 summation-order noise."""
    from cinema_amd import convvit

    views = ["sax", "lax_2c"]
    kw = dict(image_size_dict={"sax": (64, 64, 8), "lax_2c": (64, 64)}, in_chans_dict=dict.fromkeys(views, 1),
              enc_patch_size_dict={"sax": (4, 4, 1), "lax_2c": (4, 4)}, enc_scale_factor_dict={"sax": (2, 2, 1), "lax_2c": (2, 2)},
              enc_conv_chans=[64, 128], enc_conv_n_blocks=2, enc_embed_dim=256, enc_depth=2, enc_n_heads=4, dec_embed_dim=128, dec_depth=2,
              dec_n_heads=4)
    torch.manual_seed(11)
    model = CineMA(**kw).to(DEV)
    gen = torch.Generator().manual_seed(6)
    images = {v: torch.rand(3, 1, *kw["image_size_dict"][v], generator=gen).to(DEV) for v in views}
    cfg = O.MAEConfig(**kw)
    masks = {v: O.random_patch_mask(3, math.prod(cfg.grid_size(v)), 0.75, gen).to(DEV) for v in views}
    results = []
    for dense in (False, True):
        convvit.DENSE_STEM = dense
        try:
            model.zero_grad(set_to_none=True)
            loss, pred, _, _ = model(images, 0.75, enc_mask_dict=masks)
            loss.backward()
            results.append((loss.detach().float(), {k: v.detach().float() for k, v in pred.items()},
                            {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}))
        finally:
            convvit.DENSE_STEM = False
    (l0, p0, g0), (l1, p1, g1) = results
    assert abs(float(l0) - float(l1)) <= 2e-3 * abs(float(l1)), (float(l0), float(l1))  # bf16 activations, different row order in the GEMMs
    for k in p1:
        assert (p0[k] - p1[k]).abs().max() <= 2e-2 * max(1.0, float(p1[k].abs().max())), k
    assert set(g0) == set(g1)
    for k in g1:
        rel = float((g0[k] - g1[k]).norm() / (g1[k].norm() + 1e-12))
        assert rel <= 3e-2, (k, rel)  # relative L2 per parameter tensor


def test_weight_gradient_group_schedules_agree() -> None:
    """Without a gradient exchange two transformer blocks share one persistent weight-gradient launch (tape.GROUP_FLUSH_MIN = 8: whole-K tiles), with one every
    block gets its own (= 1: k-slices summed in the launch).  Same gradients up to the fp32 summation order of the k-slices, fewer launches."""
    from cinema_amd import hip as K
    from cinema_amd import tape as T

    views = ["sax", "lax_2c"]
    kw = dict(image_size_dict={"sax": (64, 64, 8), "lax_2c": (64, 64)}, in_chans_dict=dict.fromkeys(views, 1),
              enc_patch_size_dict={"sax": (4, 4, 1), "lax_2c": (4, 4)}, enc_scale_factor_dict={"sax": (2, 2, 1), "lax_2c": (2, 2)},
              enc_conv_chans=[64, 128], enc_conv_n_blocks=2, enc_embed_dim=256, enc_depth=4, enc_n_heads=4, dec_embed_dim=128, dec_depth=2,
              dec_n_heads=4)
    torch.manual_seed(12)
    model = CineMA(**kw).to(DEV)
    gen = torch.Generator().manual_seed(7)
    images = {v: torch.rand(3, 1, *kw["image_size_dict"][v], generator=gen).to(DEV) for v in views}
    cfg = O.MAEConfig(**kw)
    masks = {v: O.random_patch_mask(3, math.prod(cfg.grid_size(v)), 0.75, gen).to(DEV) for v in views}
    orig, keep = K.gemm_wgrad_grouped, (T.GROUP_FLUSH_MIN, T.GROUP_WGRAD)
    calls = {"n": 0}

    def counting(problems, **kwargs):  # noqa: ANN001, ANN003, ANN202
        calls["n"] += 1
        return orig(problems, **kwargs)

    results = []
    K.gemm_wgrad_grouped = counting
    try:
        T.GROUP_WGRAD = 2
        for flush_min in (1, 8):
            T.GROUP_FLUSH_MIN = flush_min
            calls["n"] = 0
            model.zero_grad(set_to_none=True)
            loss, _, _, _ = model(images, 0.75, enc_mask_dict=masks)
            loss.backward()
            results.append((float(loss.detach()), calls["n"], {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}))
    finally:
        K.gemm_wgrad_grouped = orig
        T.GROUP_FLUSH_MIN, T.GROUP_WGRAD = keep
    (l1, n1, g1), (l8, n8, g8) = results
    assert abs(l1 - l8) <= 1e-6 * abs(l1), (l1, l8)  # the forward pass does not depend on the schedule (its loss reduction uses fp32 atomics: last-bit noise)
    assert n8 < n1, (n1, n8)
    assert set(g1) == set(g8)
    for k in g1:
        rel = float((g1[k] - g8[k]).norm() / (g1[k].norm() + 1e-12))
        assert rel <= 1e-4, (k, rel)  # fp32 accumulation: only the order of the k-slices differs (+ last-bit noise of the loss reduction through the bf16 gradient chain)


def test_random_masks_and_api_contract() -> None:
    model = CineMA(**mini_kwargs()).to(DEV)
    images = {v: torch.rand(2, 1, *s, device=DEV) for v, s in model_sizes(model).items()}
    loss, pred, masks, metrics = model(images, 0.75)
    assert set(pred) == set(images) == set(masks)
    for v in images:
        n = model.enc_down_dict[v].patch_embed.n_patches
        assert masks[v].shape == (2, n) and masks[v].dtype == torch.bool
        assert int(masks[v].sum()) == 2 * (n - int(n * 0.25))
        assert pred[v].shape == (2, n - int(n * 0.25), math.prod(model.dec_patch_size_dict[v]))
    assert set(metrics) == {f"{v}_{m}" for v in images for m in ("target_mean", "target_std", "mse_loss")} | {"loss"}
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters() if p.requires_grad)
    sub = {"sax": images["sax"]}  # a subset of the views is allowed (mae.py:521-523)
    loss2, pred2, _, _ = model(sub, 0.5)
    assert set(pred2) == {"sax"} and torch.isfinite(loss2)
    with pytest.raises(ValueError):
        model({"bogus": images["sax"]}, 0.75)
    loss0, pred0, _, _ = model(images, 0.0)  # nothing masked: NaN loss is a value, not an exception (mae.py:604-608)
    assert math.isnan(float(loss0)) and all(p.shape[1] == 0 for p in pred0.values())


@pytest.mark.parametrize("kw", [{}, {"cross_attn": False, "norm_target": True}])
def test_recorded_step_replays_the_eager_step(kw: dict) -> None:
    """TrainStep(replay=True): forward + backward re-issued from the recorded launch list (cinema_amd/replay.py) must walk the same
    optimisation trajectory as the module code from the same seed (same random masks: identical RNG consumption), with no torch (ATen)
    device work left unaccounted inside the recorded region.  Tolerance: a few fp32 atomics (bias-gradient row sums) reorder between runs."""
    from cinema_amd.optim import TrainStep

    traj = {}
    for mode in ("eager", "replay"):
        torch.manual_seed(7)
        model = CineMA(**mini_kwargs(**kw)).to(DEV)
        step = TrainStep(model, lr=1e-3, replay=(mode == "replay"), audit=True)
        torch.manual_seed(11)
        batches = [{v: torch.rand(2, 1, *s, device=DEV) for v, s in model_sizes(model).items()} for _ in range(2)]
        out = []
        for i in range(6):
            loss, gn, metrics = step(batches[i % 2], 0.75)
            out.append((float(loss), float(gn), float(metrics["sax_mse_loss"])))
        traj[mode] = out
        if mode == "replay":
            rec = next(iter(step._recorded.values()))  # noqa: SLF001
            assert rec.unaccounted == [], rec.unaccounted[:5]
            assert rec.n_launches > 100 and len(step._recorded) == 1  # noqa: SLF001
            # a different input signature gets its own recording; update_grad=False (accumulation micro-step) leaves the weights alone
            before = step.flat.flat_param.clone()
            step({"sax": batches[0]["sax"]}, 0.5, update_grad=False)
            assert len(step._recorded) == 2 and torch.equal(before, step.flat.flat_param)  # noqa: SLF001
            step.reset_recordings()
            assert len(step._recorded) == 0  # noqa: SLF001
            loss, _, _ = step(batches[0], 0.75)  # records again
            assert len(step._recorded) == 1 and math.isfinite(float(loss))  # noqa: SLF001
    for a, b in zip(traj["eager"], traj["replay"]):
        for x, y in zip(a, b):
            assert abs(x - y) <= 2e-4 * abs(x) + 1e-6, (traj["eager"], traj["replay"])


def test_base_4view_192_first_step_vs_oracle() -> None:
    """BASELINE config 2 at its REAL shape (ViT-Base, SAX 192x192x16 + 3 LAX 192x192, 685-token encoder, 2053 x 684 cross-attention, the ragged
    80-row GEMM strips, split-tail and BK = 32 / 64 dispatch that bench.py times) at batch 2: first forward + backward of the HIP path against
    the live fp32 CPU oracle on identical weights, inputs and masks (oracle/parity.py; the same object bench.py prints as `parity`).
    Tolerances = 3 x the errors measured on an MI355X (printed by this test), see the constants."""
    import os

    from parity import NAMED_GRADS, mae_step_parity

    views = ["sax", "lax_2c", "lax_3c", "lax_4c"]
    kw = dict(image_size_dict={v: (192, 192, 16) if v == "sax" else (192, 192) for v in views}, in_chans_dict=dict.fromkeys(views, 1),
              enc_patch_size_dict={v: (4, 4, 1) if v == "sax" else (4, 4) for v in views},
              enc_scale_factor_dict={v: (2, 2, 1) if v == "sax" else (2, 2) for v in views}, enc_conv_chans=[64, 128], enc_conv_n_blocks=2,
              **get_vit_config("base"))
    torch.manual_seed(0)
    sd = CineMA(**kw).state_dict()
    par = mae_step_parity(kw, sd, batch=2, seed=7, device=DEV, threads=min(os.cpu_count() or 1, 16))
    print("config-2 parity:", {k: v for k, v in par.items() if k != "named_grads"})
    for k, v in par["named_grads"].items():
        print(f"  {k}: rel_l2 {v['rel_l2']:.4f} max {v['max_abs_over_max']:.4f}")
    assert set(par["named_grads"]) == set(NAMED_GRADS)
    assert par["loss_rel"] <= CFG2_LOSS_RTOL, par["loss_rel"]
    assert max(par["view_loss_rel"].values()) <= CFG2_VIEW_LOSS_RTOL, par["view_loss_rel"]
    assert par["grad_norm_rel"] <= CFG2_GRAD_NORM_RTOL, par["grad_norm_rel"]
    assert par["pred_max_abs"] <= PRED_ATOL, par["pred_max_abs"]
    assert par["grad_rel"] <= CFG2_NAMED_GRAD_L2, par["named_grads"]
    assert par["worst_grad_rel_l2"]["value"] <= CFG2_WORST_GRAD_L2, par["worst_grad_rel_l2"]


# fp8 path vs the ORACLE, measured on an MI355X (printed by the test), midsize 2-view model (2 + 2 blocks), batch 3:
#   mode                         loss rel   grad-norm rel   whole gradient rel-L2   worst matrix rel-L2                  worst vector rel-L2
#   bf16                         1.5e-4     1.2e-3          0.6 %                   1.3 % (decoder.blocks.1.attn.q)      1.7 % (decoder.blocks.1.norm1.bias)
#   e4m3 forward                 2.0e-3     3.3e-3          5.0 %                   12.6 %                               16.8 %
#   e4m3 forward + data grads    2.0e-3     1.5e-3          6.9 %                   18.1 %                               25.1 %
#   + e4m3 weight grads (delayed) 2.3e-3     2.0e-3          6.9 %                   16.6 %                               21.9 %
# e4m3 has 3 mantissa bits: a GEMM on e4m3 operands carries ~3-6 % relative error per output element; the worst tensors are the decoder's q projection and
# its LayerNorm (gradients that are sums of small differences of softmax terms).  Bounds = ~1.6 x measured.
FP8_LOSS_RTOL = 5e-2
FP8_GRAD_NORM_RTOL = 1e-2
FP8_MATRIX_GRAD_L2 = 0.30
FP8_VECTOR_GRAD_L2 = 0.40
FP8_WHOLE_GRAD_L2 = 0.12


def test_fp8_forward_path_vs_oracle_and_bf16() -> None:
    """BASELINE config 5's arithmetic ("fp8 MFMA path") on an MFMA-sized 2-view model: the transformer blocks' forward projections AND their data gradients on
    e4m3 operands (weights per tensor, activations / gradients per row, current scaling), weight gradients and everything else in bf16.  Loss AND GRADIENTS are
    compared with the fp32 CPU ORACLE (``oracle/parity.py::mae_fp8_grad_parity``; reference graph ``cinema/mae/mae.py:504-612``), in three modes: bf16, e4m3 forward,
    e4m3 forward + data gradients.  The flat buffers exist before the comparison, and the test asserts that the e4m3 data-gradient GEMMs really ran (their
    transposed weight shadows live in the flat buffers only).  Also: 5 recorded training steps with the flat optimiser reduce the loss."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))
    from parity import mae_fp8_grad_parity

    from cinema_amd import tape as T
    from cinema_amd.optim import TrainStep

    views = ["sax", "lax_2c"]
    kw = dict(image_size_dict={"sax": (64, 64, 8), "lax_2c": (64, 64)}, in_chans_dict=dict.fromkeys(views, 1),
              enc_patch_size_dict={"sax": (4, 4, 1), "lax_2c": (4, 4)}, enc_scale_factor_dict={"sax": (2, 2, 1), "lax_2c": (2, 2)},
              enc_conv_chans=[64, 128], enc_conv_n_blocks=1, enc_embed_dim=256, enc_depth=2, enc_n_heads=4, dec_embed_dim=128, dec_depth=2,
              dec_n_heads=4)
    torch.manual_seed(3)
    model = CineMA(**kw)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    par = mae_fp8_grad_parity(kw, sd, batch=3, seed=5, device=DEV)
    for mode in ("bf16", "fp8_forward", "fp8", "fp8_wgrad"):
        print(f"fp8 parity vs oracle [{mode}]:", par[mode])
    assert par["bf16"]["fp8_dgrad_gemms"] == 0 and par["fp8_forward"]["fp8_dgrad_gemms"] == 0
    assert par["fp8"]["fp8_dgrad_gemms"] >= 10, par["fp8"]  # q, kv, proj, fc1, fc2 of 2 + 2 blocks: the e4m3 data-gradient kernels really ran
    assert par["fp8"]["loss"] != par["bf16"]["loss"]  # ... and so did the e4m3 forward
    assert par["fp8"]["fp8_wgrad_problems"] == 0
    assert par["fp8_wgrad"]["fp8_wgrad_problems"] >= 10, par["fp8_wgrad"]  # proj, fc1, fc2 of the 2 + 2 blocks: the e4m3 weight-gradient kernel really ran
    for mode in ("fp8_forward", "fp8", "fp8_wgrad"):
        r = par[mode]
        assert r["loss_rel"] <= FP8_LOSS_RTOL, r
        assert r["grad_norm_rel"] <= FP8_GRAD_NORM_RTOL, r
        assert r["worst_matrix_rel_l2"]["value"] <= FP8_MATRIX_GRAD_L2, r
        assert r["worst_vector_rel_l2"]["value"] <= FP8_VECTOR_GRAD_L2, r
        assert r["whole_grad_rel_l2"] <= FP8_WHOLE_GRAD_L2, r
    assert par["bf16"]["worst_matrix_rel_l2"]["value"] <= 0.055 and par["bf16"]["loss_rel"] <= 2e-3, par["bf16"]
    model.to(DEV)
    g = torch.Generator().manual_seed(5)
    dimg = {v: torch.rand(3, 1, *kw["image_size_dict"][v], generator=g).to(DEV) for v in views}
    try:
        T.FP8_FORWARD = True
        step = TrainStep(model, lr=1e-3, replay=True)
        losses = [float(step(dimg, 0.75)[0]) for _ in range(5)]
        assert all(math.isfinite(v) for v in losses) and losses[-1] < losses[0], losses
        assert step.flat._fp8 is not None and step.flat._fp8["epoch"] is not None  # noqa: SLF001  (the segmented weight quantisation ran)
        assert step.flat._fp8.get("epoch_t") is not None  # noqa: SLF001  (... and the transposed copies for the data gradients)
    finally:
        T.FP8_FORWARD = False


# Full DEPTH of BASELINE config 5 (ViT-Large: 24 encoder blocks of width 1024 / 16 heads, 8 decoder blocks of width 512) at a spatial size the fp32 CPU oracle
# back-propagates in seconds (SAX 96 x 96 x 8 + one long-axis view 96 x 96, batch 2: 288 + 36 tokens per sample): the e4m3 error compounds through 24 layers,
# which the 2 + 2 block check above cannot show.  Measured on an MI355X (round 5, profiles/r05_fp8_depth.txt):
#                      loss rel   grad-norm rel   whole gradient   worst matrix (encoder.blocks.22.attn.q.weight)   matrices of one block, taken together
#   bf16               2.3e-4     9.6e-4          0.76 %           7.9 % (a tensor with ~1e-5 of the gradient norm)   0.66 - 0.82 %
#   e4m3 fwd+dgrad+wgrad 3.2e-3   1.0e-2          10.8 %           22.2 %                                              decoder 8.7 - 9.1 %, encoder 10.9 % (block 23) ... 12.0 % (block 0)
# i.e. the error grows by a tenth over 24 layers - it does not compound.  Bounds = 1.6 x measured.
FP8_DEPTH_WHOLE_GRAD_L2 = 0.175
FP8_DEPTH_MATRIX_GRAD_L2 = 0.36
FP8_DEPTH_BLOCK_L2 = 0.195


def large_depth_kwargs() -> dict:
    from cinema_amd.vit import get_vit_config

    views = ["sax", "lax_2c"]
    return dict(image_size_dict={"sax": (96, 96, 8), "lax_2c": (96, 96)}, in_chans_dict=dict.fromkeys(views, 1), enc_patch_size_dict={"sax": (4, 4, 1), "lax_2c": (4, 4)},
                enc_scale_factor_dict={"sax": (2, 2, 1), "lax_2c": (2, 2)}, enc_conv_chans=[64, 128], enc_conv_n_blocks=2, **get_vit_config("large"))


def test_fp8_gradients_at_vit_large_depth_vs_oracle() -> None:
    """Gradients of the full fp8 path (e4m3 forward, data-gradient and weight-gradient GEMMs, per-tensor delayed scaling) against the fp32 CPU oracle on a
    model with config 5's FULL depth and widths (24 + 8 blocks, 1024 / 512 channels; reference graph ``cinema/mae/mae.py:504-612`` at
    ``cinema/vit.py:784-831`` "large") and a small spatial size; the bf16 path on the same model beside it.  Reported per mode: loss, gradient norm, whole
    gradient, worst matrix, and the error of every block's matrices taken together (``block_matrix_rel_l2``: growth with depth)."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))
    from parity import mae_fp8_grad_parity

    kw = large_depth_kwargs()
    torch.manual_seed(11)
    sd = {k: v.detach().clone() for k, v in CineMA(**kw).state_dict().items()}
    par = mae_fp8_grad_parity(kw, sd, batch=2, seed=13, device=DEV, modes=("bf16", "fp8_wgrad"))
    for mode in ("bf16", "fp8_wgrad"):
        print(f"fp8 depth parity vs oracle [{mode}]:", par[mode])
    b16, f8 = par["bf16"], par["fp8_wgrad"]
    assert f8["fp8_dgrad_gemms"] >= 5 * 32 - 8 and f8["fp8_wgrad_problems"] >= 3 * 32, f8  # the e4m3 kernels really ran in all 32 blocks
    assert b16["loss_rel"] <= 2e-3 and b16["grad_norm_rel"] <= 1e-2 and b16["whole_grad_rel_l2"] <= 0.03, b16
    # round 6: the worst bf16 matrix was encoder.blocks.22.attn.q.weight at 7.9 % - the bf16 rounding of the attention OUTPUT inside delta = rowsum(dO O), amplified by
    # the keys' common component (profiles/r06_i_attn_dq_error.txt); with both halves of O it is a decoder q weight at 1.5 %.  Required: 3 %.
    assert b16["worst_matrix_rel_l2"]["value"] <= 3e-2, b16["worst_matrix_rel_l2"]
    assert f8["loss_rel"] <= FP8_LOSS_RTOL and f8["grad_norm_rel"] <= 5e-2, f8
    assert f8["whole_grad_rel_l2"] <= FP8_DEPTH_WHOLE_GRAD_L2, f8
    assert f8["worst_matrix_rel_l2"]["value"] <= FP8_DEPTH_MATRIX_GRAD_L2, f8
    assert max(f8["block_matrix_rel_l2"].values()) <= FP8_DEPTH_BLOCK_L2, f8


def test_fp8_300_step_trajectory_tracks_bf16_at_full_depth() -> None:
    """Does the fp8 path TRAIN at config 5's depth?  300 optimisation steps (forward, backward, clip, AdamW, recorded step) of the ViT-Large-depth model (24 + 8
    blocks, 1024 / 512 channels, SAX 96 x 96 x 8 + one long-axis view, batch 2) on a fixed set of 8 batches with per-step random masks drawn from one seeded stream,
    once in bf16 and once with e4m3 forward + data-gradient + weight-gradient GEMMs from the same initial weights.  A single gradient of the fp8 path is 11 % off the
    fp32 oracle's (9 points of that from the e4m3 FORWARD alone: profiles/r06_j_fp8_depth_modes.txt) - e4m3's 3-bit mantissa, not a scaling artefact (per-row scales
    give the same number).  What the trajectory shows is that this noise averages out: the loss curves stay together.  Required: both runs reduce the loss by more
    than a third, the fp8 curve's mean over the last 50 steps within 5 % of bf16's, and within 10 % over every window of 25 steps after step 50."""
    from cinema_amd import tape as T
    from cinema_amd.optim import TrainStep

    kw = large_depth_kwargs()
    views = list(kw["image_size_dict"])
    gen = torch.Generator().manual_seed(21)
    batches = [{v: torch.rand(2, 1, *kw["image_size_dict"][v], generator=gen).to(DEV) for v in views} for _ in range(8)]
    torch.manual_seed(11)
    sd = {k: v.detach().clone() for k, v in CineMA(**kw).state_dict().items()}
    curves = {}
    prev = (T.FP8_FORWARD, T.FP8_DGRAD, T.FP8_WGRAD)
    try:
        for name, fp8 in (("bf16", False), ("fp8", True)):
            T.FP8_FORWARD, T.FP8_DGRAD, T.FP8_WGRAD = fp8, True, True
            model = CineMA(**kw)
            model.load_state_dict(sd)
            model.to(DEV)
            step = TrainStep(model, lr=2e-4, betas=(0.9, 0.95), weight_decay=0.05, clip_grad=5.0, replay=True)
            torch.manual_seed(33)  # the same stream of mask noise for both runs
            losses = []
            for i in range(300):
                loss, _, _ = step(batches[i % 8], 0.75)
                losses.append(loss.detach().float().reshape(()).clone())  # (a replayed step hands back the same buffer every time)
            curves[name] = torch.stack(losses).cpu()
            del step, model
            torch.cuda.empty_cache()
    finally:
        T.FP8_FORWARD, T.FP8_DGRAD, T.FP8_WGRAD = prev
    b, f = curves["bf16"], curves["fp8"]
    assert bool(torch.isfinite(b).all()) and bool(torch.isfinite(f).all())
    head_b, tail_b, tail_f = float(b[:10].mean()), float(b[-50:].mean()), float(f[-50:].mean())
    windows = [(float(f[i:i + 25].mean()), float(b[i:i + 25].mean())) for i in range(50, 300, 25)]
    worst = max(abs(x - y) / y for x, y in windows)
    print(f"300-step trajectory at ViT-Large depth: bf16 {head_b:.4f} -> {tail_b:.4f}, fp8 {float(f[:10].mean()):.4f} -> {tail_f:.4f}; last-50 mean rel {abs(tail_f - tail_b) / tail_b:.4f}, "
          f"worst 25-step window rel {worst:.4f}")
    assert tail_b <= head_b * 0.67 and tail_f <= float(f[:10].mean()) * 0.67, (head_b, tail_b, tail_f)
    assert abs(tail_f - tail_b) <= 0.05 * tail_b, (tail_f, tail_b)
    assert worst <= 0.10, windows


def test_fp8_training_trajectory_vs_oracle() -> None:
    """Six optimisation steps (forward, backward, clip, AdamW; identical injected masks and inputs per step) of the full fp8 path - e4m3 forward, data-gradient
    AND weight-gradient GEMMs, per-tensor delayed scaling (the first step records the maxima and runs the per-row / bf16 forms) - against the fp32 CPU oracle's
    ``Trainer`` from the same initial weights: every step's loss within 5e-2 of the oracle's (SURVEY 8d's fp8 tolerance; measured on an MI355X <= 6e-3), the
    pre-clip gradient norm within 5e-2 (measured <= 1.2e-2), and the e4m3 weight-gradient kernel ran in every step after the first."""
    from cinema_amd import hip as K
    from cinema_amd import tape as T
    from cinema_amd.optim import TrainStep

    views = ["sax", "lax_2c"]
    kw = dict(image_size_dict={"sax": (64, 64, 8), "lax_2c": (64, 64)}, in_chans_dict=dict.fromkeys(views, 1),
              enc_patch_size_dict={"sax": (4, 4, 1), "lax_2c": (4, 4)}, enc_scale_factor_dict={"sax": (2, 2, 1), "lax_2c": (2, 2)},
              enc_conv_chans=[64, 128], enc_conv_n_blocks=1, enc_embed_dim=256, enc_depth=2, enc_n_heads=4, dec_embed_dim=128, dec_depth=2,
              dec_n_heads=4)
    torch.manual_seed(11)
    model = CineMA(**kw)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = O.MAEConfig(**kw)
    trainer = O.Trainer(sd, cfg, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05, clip_grad=5.0)
    gen = torch.Generator().manual_seed(12)
    steps = []
    for _ in range(6):
        images = {v: torch.rand(3, 1, *kw["image_size_dict"][v], generator=gen) for v in views}
        masks = {v: O.random_patch_mask(3, math.prod(cfg.grid_size(v)), 0.75, gen) for v in views}
        steps.append((images, masks))
    ref = [trainer.step(im, mk)[:2] for im, mk in steps]
    model.to(DEV)
    calls = {"n": 0}
    orig = K.gemm_fp8_wgrad_grouped

    def counting(problems):  # noqa: ANN001, ANN202
        calls["n"] += 1
        return orig(problems)

    prev = (T.FP8_FORWARD, T.FP8_DGRAD, T.FP8_WGRAD)
    K.gemm_fp8_wgrad_grouped = counting
    try:
        T.FP8_FORWARD, T.FP8_DGRAD, T.FP8_WGRAD = True, True, True
        step = TrainStep(model, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05, clip_grad=5.0)
        got, per_step = [], []
        for im, mk in steps:
            before = calls["n"]
            loss, gn, _ = step({k: v.to(DEV) for k, v in im.items()}, 0.75, enc_mask_dict={k: v.to(DEV) for k, v in mk.items()})
            got.append((float(loss), float(gn)))
            per_step.append(calls["n"] - before)
    finally:
        K.gemm_fp8_wgrad_grouped = orig
        T.FP8_FORWARD, T.FP8_DGRAD, T.FP8_WGRAD = prev
    rel = [(abs(g[0] - float(r[0])) / float(r[0]), abs(g[1] - float(r[1])) / float(r[1])) for g, r in zip(got, ref)]
    print("fp8 trajectory vs oracle (loss rel, grad-norm rel) per step:", [(round(a, 5), round(b, 5)) for a, b in rel], "fp8 weight-gradient launches per step:", per_step)
    # grouped launches from the second step on: without a gradient exchange two blocks share one (tape.GROUP_FLUSH_MIN), i.e. one per encoder / decoder pair here
    assert per_step[0] == 0 and all(n >= 2 for n in per_step[1:]), per_step
    assert max(a for a, _ in rel) <= 5e-2 and max(b for _, b in rel) <= 5e-2, rel
    assert got[-1][0] < got[0][0]


def test_fp8_replay_of_a_second_model_calibrates_before_it_records() -> None:
    """Round-6 fault: the delayed-scaling registry is per device, so after one fp8 model had trained in the process a SECOND model's first replayed step was taken as
    "calibrated", recorded straight away and replayed launches whose descriptor tensors had meanwhile been replaced (memory fault in
    cinema_quantize_fp8_segments_t).  Required: with the first model still alive, the second model's first step is eager (no recording yet), the recording is
    taken on its second step, its launch list holds each joint re-quantisation once, and 12 replayed steps with small allocations in between stay finite and
    reduce the loss."""
    from cinema_amd import hip as K
    from cinema_amd import tape as T
    from cinema_amd.optim import TrainStep

    views = ["sax", "lax_2c"]
    kw = dict(image_size_dict={"sax": (64, 64, 8), "lax_2c": (64, 64)}, in_chans_dict=dict.fromkeys(views, 1),
              enc_patch_size_dict={"sax": (4, 4, 1), "lax_2c": (4, 4)}, enc_scale_factor_dict={"sax": (2, 2, 1), "lax_2c": (2, 2)},
              enc_conv_chans=[64, 128], enc_conv_n_blocks=1, enc_embed_dim=256, enc_depth=3, enc_n_heads=4, dec_embed_dim=128, dec_depth=2,
              dec_n_heads=4)
    gen = torch.Generator().manual_seed(5)
    batches = [{v: torch.rand(2, 1, *kw["image_size_dict"][v], generator=gen).to(DEV) for v in views} for _ in range(4)]
    prev = (T.FP8_FORWARD, T.FP8_DGRAD, T.FP8_WGRAD)
    try:
        T.FP8_FORWARD, T.FP8_DGRAD, T.FP8_WGRAD = True, True, True
        torch.manual_seed(3)
        first = CineMA(**kw).to(DEV)
        step_a = TrainStep(first, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05, clip_grad=5.0)
        for i in range(2):
            step_a(batches[i], 0.75)
        assert not T.fp8_calibrating()  # the registry alone would now wave a new model through
        torch.manual_seed(4)
        second = CineMA(**kw).to(DEV)
        step_b = TrainStep(second, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05, clip_grad=5.0, replay=True)
        losses = [float(step_b(batches[0], 0.75)[0])]
        assert not step_b._recorded, "the first fp8 step of a model must run eagerly"  # noqa: SLF001
        keep = []
        for i in range(1, 13):
            loss = step_b(batches[i % 4], 0.75)[0]
            keep.append(loss.detach().float().reshape(()).clone())  # small allocations between the steps: what landed on the freed descriptors in round 6
            losses.append(float(loss))
        (rec,) = step_b._recorded.values()  # noqa: SLF001
        names = [fn.__name__ for fn, _ in rec.calls if fn is not None]
        assert names.count("cinema_quantize_fp8_segments_t") <= 2, names.count("cinema_quantize_fp8_segments_t")  # per-weight + joint shadows, once each
        del step_a, first
    finally:
        T.FP8_FORWARD, T.FP8_DGRAD, T.FP8_WGRAD = prev
    assert all(math.isfinite(x) for x in losses) and losses[-1] < losses[0], losses


def test_large_config_256_fp8_first_step_loss_vs_oracle() -> None:
    """BASELINE config 5 at its OWN shape (ViT-Large, SAX 256 x 256 x 24 + 3 LAX 256 x 256, batch 1): first-step loss of the fp8 path (e4m3 forward
    projections) and of the bf16 path against the fp32 CPU oracle's forward on identical weights, inputs and masks (oracle/parity.py::mae_loss_parity).
    Stated tolerance: fp8 loss rel <= 5e-2, bf16 <= 2e-2; the fp8 loss must differ from the bf16 one (the e4m3 kernels really ran)."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))
    from parity import mae_loss_parity

    from cinema_amd.vit import get_vit_config

    views = ["sax", "lax_2c", "lax_3c", "lax_4c"]
    kw = dict(image_size_dict={v: (256, 256, 24) if v == "sax" else (256, 256) for v in views}, in_chans_dict=dict.fromkeys(views, 1),
              enc_patch_size_dict={v: (4, 4, 1) if v == "sax" else (4, 4) for v in views},
              enc_scale_factor_dict={v: (2, 2, 1) if v == "sax" else (2, 2) for v in views}, enc_conv_chans=[64, 128], enc_conv_n_blocks=2,
              **get_vit_config("large"))
    torch.manual_seed(3)
    sd = {k: v.detach().clone() for k, v in CineMA(**kw).state_dict().items()}
    p8 = mae_loss_parity(kw, sd, batch=1, device=DEV, fp8=True, threads=16)
    p16 = mae_loss_parity(kw, sd, batch=1, device=DEV, fp8=False, threads=16)
    print("config-5 shape, first-step loss vs oracle:", p8, p16)
    assert p8["loss_rel"] <= 5e-2 and max(p8["view_loss_rel"].values()) <= 5e-2, p8
    assert p16["loss_rel"] <= 2e-2, p16
    assert p8["loss"] != p16["loss"]


CFG5_NAMED = ("encoder.blocks.0.attn.kv.weight", "encoder.blocks.23.mlp.fc1.weight", "decoder.blocks.0.attn.q.weight", "decoder.blocks.7.mlp.fc2.weight",
              "enc_down_dict.sax.conv_blocks.0.conv.0.mlp.fc1.weight", "pred_head_dict.sax.weight")


def test_large_config_256_gradients_vs_oracle_at_its_own_shape() -> None:
    """BASELINE config 5 at its OWN spatial size and depth (ViT-Large 24 + 8 blocks, SAX 256 x 256 x 24 + 3 LAX 256 x 256, 6912 tokens, batch 1): loss AND gradients
    against the fp32 CPU oracle (its forward + backward at this shape takes 13 s on the GPU box's host), bf16 path and full e4m3 path, with six named tensors from
    both ends of the encoder, the decoder, the stem and a head.  bf16: REQUIRED whole gradient <= 2 %, every matrix <= 3 % (measured 0.87 % / 1.6 %).  e4m3: the
    review's 5 % / 10 % cannot be met by 3-bit-mantissa operands (the e4m3 FORWARD alone costs 9.3 % of the gradient at this depth, per-row scales change nothing:
    profiles/r06_j_fp8_depth_modes.txt); what is asserted is a regression guard from a noise model, not a requirement: an e4m3 operand carries an rms relative
    rounding error of 2^-4 / sqrt 3 = 3.6 %, a product of two 5.1 %, and the shortest loop from a weight back to itself passes five such products (fc1, fc2 forward,
    two data gradients, the weight gradient): sqrt 5 x 5.1 % = 11.4 %; guard = 1.3 x that for the whole gradient and for the named tensors, 30 % for the worst
    matrix (measured 12.1 % / 10-14 % / 25.9 %).  Whether that noise harms training is the trajectory test's question."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))
    from parity import mae_fp8_grad_parity

    from cinema_amd.vit import get_vit_config

    views = ["sax", "lax_2c", "lax_3c", "lax_4c"]
    kw = dict(image_size_dict={v: (256, 256, 24) if v == "sax" else (256, 256) for v in views}, in_chans_dict=dict.fromkeys(views, 1),
              enc_patch_size_dict={v: (4, 4, 1) if v == "sax" else (4, 4) for v in views},
              enc_scale_factor_dict={v: (2, 2, 1) if v == "sax" else (2, 2) for v in views}, enc_conv_chans=[64, 128], enc_conv_n_blocks=2,
              **get_vit_config("large"))
    torch.manual_seed(0)
    sd = {k: v.detach().clone() for k, v in CineMA(**kw).state_dict().items()}
    par = mae_fp8_grad_parity(kw, sd, batch=1, seed=17, device=DEV, threads=16, modes=("bf16", "fp8_wgrad"), report=CFG5_NAMED)
    b16, f8 = par["bf16"], par["fp8_wgrad"]
    print("config-5 own shape vs oracle:", {m: {k: par[m][k] for k in ("loss_rel", "grad_norm_rel", "whole_grad_rel_l2", "worst_matrix_rel_l2", "named_rel_l2")} for m in ("bf16", "fp8_wgrad")})
    assert set(b16["named_rel_l2"]) == set(CFG5_NAMED)
    assert b16["loss_rel"] <= 2e-3 and b16["whole_grad_rel_l2"] <= 2e-2 and b16["worst_matrix_rel_l2"]["value"] <= 3e-2 and max(b16["named_rel_l2"].values()) <= 2e-2, b16
    assert f8["fp8_dgrad_gemms"] >= 5 * 32 - 8 and f8["fp8_wgrad_problems"] >= 3 * 32, f8
    guard = 1.3 * (5 ** 0.5) * (2 ** 0.5) * (2.0 ** -4 / 3 ** 0.5)
    assert f8["loss_rel"] <= FP8_LOSS_RTOL and f8["whole_grad_rel_l2"] <= guard and max(f8["named_rel_l2"].values()) <= guard and f8["worst_matrix_rel_l2"]["value"] <= 0.30, f8


def test_large_config_256_step_properties() -> None:
    """BASELINE config 5 shape (ViT-Large, 4 views, SAX 256x256x24 + LAX 256x256, 6144 + 3 x 256 tokens) at batch 1: too large for the CPU
    oracle inside a test, so size-independent properties are checked instead: mask counts, finite loss / gradient norm, a loss that falls on
    a fixed batch, and the recorded (replayed) steps continuing the eager trajectory (bf16 compute; the fp8 path at this shape is checked against the
    oracle by ``test_large_config_256_fp8_first_step_loss_vs_oracle``)."""
    from cinema_amd.optim import TrainStep
    from cinema_amd.vit import get_vit_config

    views = ["sax", "lax_2c", "lax_3c", "lax_4c"]
    kw = dict(image_size_dict={v: (256, 256, 24) if v == "sax" else (256, 256) for v in views}, in_chans_dict=dict.fromkeys(views, 1),
              enc_patch_size_dict={v: (4, 4, 1) if v == "sax" else (4, 4) for v in views},
              enc_scale_factor_dict={v: (2, 2, 1) if v == "sax" else (2, 2) for v in views}, enc_conv_chans=[64, 128], enc_conv_n_blocks=2,
              **get_vit_config("large"))
    torch.manual_seed(3)
    model = CineMA(**kw).to(DEV)
    g = torch.Generator().manual_seed(4)
    batch = {v: torch.rand(1, 1, *s, generator=g).to(DEV) for v, s in kw["image_size_dict"].items()}
    loss, pred, masks, _ = model(batch, 0.75)
    assert masks["sax"].shape == (1, 6144) and int(masks["sax"].sum()) == 6144 - 1536 and pred["sax"].shape == (1, 4608, 256)
    assert masks["lax_2c"].shape == (1, 256) and int(masks["lax_2c"].sum()) == 192
    step = TrainStep(model, lr=1e-4)
    losses = []
    for i in range(6):
        step.replay = i >= 2  # two eager steps, then one recording step and three replays
        loss, gn, _ = step(batch, 0.75)
        losses.append(float(loss))
        assert math.isfinite(losses[-1]) and math.isfinite(float(gn))
    assert losses[-1] < 0.7 * losses[0], losses


def model_sizes(model: CineMA) -> dict:
    out = {}
    for v in model.views:
        enc = model.enc_down_dict[v]
        out[v] = tuple(g * p for g, p in zip(enc.patch_embed.grid_size, enc.eff_patch_size))
    return out


def test_cpu_tensor_fails_loudly() -> None:
    from cinema_amd.hip import HipLibraryError

    model = CineMA(**mini_kwargs())  # parameters on the CPU
    with pytest.raises(HipLibraryError):
        model({"sax": torch.rand(1, 1, 32, 32, 4)}, 0.75)


def test_three_step_optimisation_trajectory_vs_reference_golden() -> None:
    """cinema_amd.optim.TrainStep (fused clip + AdamW on flat buffers, 2-group weight decay, LR schedule) against 3 steps of the
    reference harness (GradScaler + adjust_learning_rate + AdamW) captured on cfg 1.  Adam's first updates are ~lr * sign(g): the
    parameter UPDATES (p - p_init) are compared by relative L2 error and sign agreement.  Bounds = ~3 x measured on an MI355X (loss rel
    2.3e-4, grad-norm rel 4.6e-3, update rel-L2 3.0 %, sign agreement 99.7 %, mean-abs deviation 0.006 x cumulative lr)."""
    from cinema_amd.optim import TrainStep, adjust_learning_rate

    g = load_golden("tiny_sax_trajectory.safetensors")
    init = split(load_golden("tiny_sax.safetensors"), "param/")
    model = CineMA(**tiny_kwargs())
    model.load_state_dict(init)
    model.to(DEV)
    step = TrainStep(model, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05, clip_grad=5.0)
    gen = torch.Generator().manual_seed(7)
    named = dict(model.named_parameters())
    lr_sum = 0.0
    for i in range(3):
        lr = adjust_learning_rate(step.optimizer, i / 4, 1, 5, 1e-3, 1e-6)
        assert lr == pytest.approx(float(g[f"step{i}/lr"][0]), rel=1e-12, abs=1e-15)
        lr_sum += lr
        image = torch.rand(2, 1, 128, 128, 8, generator=gen)
        assert torch.equal(image.flatten()[:64], g[f"step{i}/image_head"])
        loss, norm, _ = step({"sax": image.to(DEV)}, 0.75, enc_mask_dict={"sax": g[f"step{i}/mask"].bool().to(DEV)})
        assert abs(float(loss) - float(g[f"step{i}/loss"][0])) <= 1e-3 * float(g[f"step{i}/loss"][0]), (i, float(loss))
        assert abs(float(norm) - float(g[f"step{i}/grad_norm"][0])) <= 1.5e-2 * float(g[f"step{i}/grad_norm"][0]), (i, float(norm))
        for k, t in split(g, f"step{i}/param/").items():
            got = named[k].detach().float().cpu()
            dev = (got - t).abs().mean()
            d_ref, d_got = t - init[k], got - init[k]
            print(f"trajectory step {i} {k}: loss rel {abs(float(loss) - float(g[f'step{i}/loss'][0])) / float(g[f'step{i}/loss'][0]):.2e} norm rel "
                  f"{abs(float(norm) - float(g[f'step{i}/grad_norm'][0])) / float(g[f'step{i}/grad_norm'][0]):.2e} mean-abs dev / lr_sum {float(dev) / max(lr_sum, 1e-30):.3f} "
                  f"update rel-L2 {float((d_got - d_ref).norm() / d_ref.norm().clamp_min(1e-30)):.3f} sign agreement {float((torch.sign(d_got) == torch.sign(d_ref)).float().mean()):.4f}")
            assert float(dev) <= 0.02 * lr_sum + 1e-7, (i, k, float(dev), lr_sum)
            if float(d_ref.norm()) > 0:
                assert float((d_got - d_ref).norm() / d_ref.norm()) <= 0.1, (i, k)
                assert float((torch.sign(d_got) == torch.sign(d_ref)).float().mean()) >= 0.99, (i, k)


# ------------------------------------------------------------------------------------------------ DownsampleEncoder
@pytest.mark.parametrize("nd", [2, 3])
def test_downsample_encoder_vs_reference_golden_including_pos_embed_interpolation(nd: int) -> None:
    """``DownsampleEncoder.forward`` through the HIP path against the reference layer goldens: the built grid with a stem mask (skips + tokens) and an
    input whose grid differs from the built one, which takes the pos-embed interpolation branch (bicubic 2-D / trilinear 3-D, reference
    ``convvit.py:140-163``).  bf16 MFMA operands: max-abs 5e-2 on O(1) activations."""
    from cinema_amd.convvit import DownsampleEncoder

    g = load_golden("layers.safetensors")
    size = (32, 32) if nd == 2 else (32, 32, 4)
    enc = DownsampleEncoder(image_size=size, in_chans=1, patch_size=(4, 4, 1)[:nd], scale_factor=(2, 2, 1)[:nd], conv_chans=[8, 16], conv_n_blocks=1,
                            embed_dim=24, norm="layer")
    enc.load_state_dict(split(g, f"down{nd}d/param/"))
    enc.to(DEV)
    skips, tok = enc(g[f"down{nd}d/image"].to(DEV), g[f"down{nd}d/mask"].bool().to(DEV))
    assert tok.shape == g[f"down{nd}d/tokens"].shape and (tok.float().cpu() - g[f"down{nd}d/tokens"]).abs().max() <= 5e-2
    for i, s in enumerate(skips):  # this entry point evaluates every voxel like the reference (the mask only zeroes the depthwise convolution's input)
        ref = g[f"down{nd}d/skip{i}"]
        assert s.shape == ref.shape and (s.float().cpu() - ref).abs().max() <= 5e-2, (i, float((s.float().cpu() - ref).abs().max()))
    other = g[f"down{nd}d/image_other"].to(DEV)
    _, tok2 = enc(other, None)
    ref2 = g[f"down{nd}d/tokens_other"]
    assert tok2.shape == ref2.shape and tok2.shape[1] != tok.shape[1]
    assert (tok2.float().cpu() - ref2).abs().max() <= 5e-2, float((tok2.float().cpu() - ref2).abs().max())
    _, tok3 = enc(other, None)  # second call: the resampled table comes from the cache
    assert torch.equal(tok2, tok3)


# ------------------------------------------------------------------------------------------------ ConvViT
def _convvit_model():  # noqa: ANN202
    import json

    from cinema_amd.convvit import ConvViT
    from conftest import GOLDEN

    kw = json.loads((GOLDEN / "convvit_meta.json").read_text())["kwargs"]
    for key in ("image_size_dict", "enc_patch_size_dict", "enc_scale_factor_dict"):
        kw[key] = {v: tuple(s) for v, s in kw[key].items()}
    g = load_golden("convvit_mini.safetensors")
    model = ConvViT(**kw)
    model.load_state_dict(split(g, "param/"))
    return model.to(DEV), g


def test_convvit_logits_and_features_vs_reference_golden() -> None:
    """``ConvViT.feature_forward`` / ``forward`` (all reduce modes, with and without stem masks, 2 frames per view) against the reference.
    Tolerances: bf16 MFMA operands, fp32 accumulation - features (LayerNorm-normalised, O(1)) max-abs 5e-2, logits max-abs 3e-2."""
    model, g = _convvit_model()
    images = {k: v.to(DEV) for k, v in split(g, "image/").items()}
    masks = {k: v.bool().to(DEV) for k, v in split(g, "mask/").items()}
    for tag, md in (("nomask", None), ("mask", masks)):
        feats = model.feature_forward(images, md)
        for k, t in split(g, f"feature_{tag}/").items():
            assert feats[k].shape == t.shape, (tag, k)
            assert (feats[k].float().cpu() - t).abs().max() <= 5e-2, (tag, k)
        for reduce, t in split(g, f"logits_{tag}/").items():
            out = model(images, md, reduce=reduce)
            assert out.shape == t.shape, (tag, reduce)
            assert (out.float().cpu() - t).abs().max() <= 3e-2, (tag, reduce, float((out.float().cpu() - t).abs().max()))


def test_convvit_gradients_vs_reference_golden() -> None:
    model, g = _convvit_model()
    images = {k: v.to(DEV) for k, v in split(g, "image/").items()}
    masks = {k: v.bool().to(DEV) for k, v in split(g, "mask/").items()}
    (model(images, masks, reduce="all") * g["grad/coef"].to(DEV)).sum().backward()
    named = dict(model.named_parameters())
    for k, t in split(g, "grad/").items():
        if k == "coef":
            continue
        got = named[k].grad.float().cpu()
        rel = float((got - t).norm() / (t.norm() + 1e-12))
        assert rel <= 6e-2, (k, rel)  # relative L2, same bound as the MAE gradient checks


# ------------------------------------------------------------------------------------------------ ConvUNetR
def test_conv_res_block_and_upsample_decoder_vs_reference_golden() -> None:
    """Layer KATs through the HIP path: ConvResBlock with a channel change (5 -> 16 channels, 3-D: non-multiple-of-8 im2col rows and the
    1x1 shortcut) and a 2-level UpsampleDecoder (transposed conv + skip add).  bf16 MFMA operands: max-abs 4e-2 on O(1) outputs."""
    from cinema_amd import tape as T
    from cinema_amd.conv import ConvResBlock, Volume
    from cinema_amd.segmentation.convunetr import UpsampleDecoder

    g = load_golden("convunetr_mini.safetensors")
    blk = ConvResBlock(n_dims=3, in_chans=5, out_chans=16, norm="layer")
    blk.load_state_dict(split(g, "resblock/param/"))
    y = blk.to(DEV)(g["resblock/x"].to(DEV))
    assert y.shape == g["resblock/y"].shape and (y.float().cpu() - g["resblock/y"]).abs().max() <= 4e-2
    dec = UpsampleDecoder(n_dims=2, chans=(8, 16), patch_size=(2, 2), scale_factor=(2, 2), norm="layer")
    dec.load_state_dict(split(g, "updec/param/"))
    dec.to(DEV)
    e0, e2 = g["updec/e0"].to(DEV), g["updec/e2"].to(DEV)

    def run(tp, a, b):  # noqa: ANN001, ANN202
        return [dec.tape_forward(tp, [Volume(a, 1, (8, 8), 8), None, Volume(b, 1, (2, 2), 16)]).var], []

    (out,) = T.taped_call(run, [e0.movedim(1, -1).reshape(-1, 8).contiguous(), e2.movedim(1, -1).reshape(-1, 16).contiguous()], list(dec.parameters()))
    out = out.reshape(1, 8, 8, -1).movedim(-1, 1)
    assert (out.float().cpu() - g["updec/y"]).abs().max() <= 4e-2


def test_convunetr_mfma_shapes_vs_reference_golden() -> None:
    """ConvUNetR with the ACDC decoder widths (32 .. 512 channels: every implicit-GEMM / z-blocked convolution form of the real config) on a
    64 x 64 x 4 volume, E = 256 / head_dim 64: logits and gradients against tests/golden/convunetr_mid.safetensors, written by the upstream
    reference (oracle/make_golden_convunetr.py::gen_mid); weights from the seeded construction (fingerprint checked).  Logits max-abs <= 3 % of the
    logit range, gradient norm rel <= 2e-2, gradients by tensor class (matrices 3 %, vectors 6 % relative L2)."""
    import json

    from cinema_amd.segmentation.convunetr import ConvUNetR
    from conftest import GOLDEN

    g = load_golden("convunetr_mid.safetensors")
    meta = json.loads((GOLDEN / "convunetr_mid_meta.json").read_text())
    kw = dict(image_size_dict={"sax": (64, 64, 4)}, in_chans_dict={"sax": 1}, out_chans=4, enc_patch_size_dict={"sax": (4, 4, 1)},
              enc_scale_factor_dict={"sax": (2, 2, 1)}, enc_conv_chans=[64, 128], enc_conv_n_blocks=1, enc_embed_dim=256, enc_depth=2, enc_n_heads=4,
              dec_chans=(32, 64, 128, 256, 512), dec_patch_size_dict={"sax": (2, 2, 1)}, dec_scale_factor_dict={"sax": (2, 2, 1)})
    torch.manual_seed(meta["seed_init"])
    model = ConvUNetR(**kw)
    fingerprint_ok(model.state_dict(), meta["params"])
    model.to(DEV).eval()
    logits = model({"sax": g["image/sax"].to(DEV)})["sax"]
    ref = g["logits/sax"]
    err = float((logits.float().cpu() - ref).abs().max())
    assert err <= 3e-2 * float(ref.abs().max()), (err, float(ref.abs().max()))
    (logits * g["coef/sax"].to(DEV)).sum().backward()
    named = dict(model.named_parameters())
    gn = math.sqrt(sum(float(p.grad.double().pow(2).sum()) for p in model.parameters() if p.grad is not None))
    gn_ref = math.sqrt(float(g["grad_sq_norm"][0]))
    assert abs(gn - gn_ref) <= 2e-2 * gn_ref, (gn, gn_ref)
    worst = check_grads_by_class(named, split(g, "grad/"), meta)
    print("ConvUNetR mid vs REFERENCE golden: logits max-abs", err, "of", float(ref.abs().max()), "grad norm rel", abs(gn - gn_ref) / gn_ref, "worst by class", worst)


def test_convunetr_logits_and_gradients_vs_reference_golden() -> None:
    import json

    from cinema_amd.segmentation.convunetr import ConvUNetR
    from conftest import GOLDEN

    kw = json.loads((GOLDEN / "convunetr_meta.json").read_text())["kwargs"]
    for key in ("image_size_dict", "enc_patch_size_dict", "enc_scale_factor_dict", "dec_patch_size_dict", "dec_scale_factor_dict"):
        kw[key] = {v: tuple(s) for v, s in kw[key].items()}
    kw["dec_chans"] = tuple(kw["dec_chans"])
    g = load_golden("convunetr_mini.safetensors")
    model = ConvUNetR(**kw)
    model.load_state_dict({k: v for k, v in split(g, "param/").items() if not k.startswith(("resblock", "updec"))})
    model.to(DEV).eval()
    images = {k: v.to(DEV) for k, v in split(g, "image/").items()}
    logits = model(images)
    for v, t in split(g, "logits/").items():
        assert logits[v].shape == t.shape, v
        err = float((logits[v].float().cpu() - t).abs().max())
        assert err <= 5e-2 * max(1.0, float(t.abs().max())), (v, err)  # bf16 MFMA operands through ~30 conv / GEMM layers
    sum((logits[v] * g[f"coef/{v}"].to(DEV)).sum() for v in images).backward()
    named = dict(model.named_parameters())
    for k, t in split(g, "grad/").items():
        got = named[k].grad.float().cpu()
        rel = float((got - t).norm() / (t.norm() + 1e-12))
        assert rel <= 8e-2, (k, rel)  # relative L2 per tensor (a deeper chain than the MAE checks: 6e-2 there)
    # the step-level API of the reference (cinema/segmentation/train.py:106-146): batch keys {view}_image / {view}_label, metric keys as it builds them
    from cinema_amd.segmentation.train import segmentation_loss

    model.zero_grad(set_to_none=True)
    views = list(images)
    batch = {f"{v}_image": images[v] for v in views}
    gen = torch.Generator().manual_seed(2)
    batch.update({f"{v}_label": torch.randint(0, logits[v].shape[1], (logits[v].shape[0], 1, *logits[v].shape[2:]), generator=gen) for v in views})
    loss, metrics = segmentation_loss(model, batch, views, torch.device(DEV))
    assert loss.dim() == 0 and torch.isfinite(loss) and abs(metrics["loss"] - float(loss)) < 1e-6
    expected = {"loss", "cross_entropy", "mean_dice_loss"} | {f"{v}_{k}" for v in views for k in ("cross_entropy", "mean_dice_loss", "loss", f"{v}_loss")}
    assert set(metrics) == expected, set(metrics) ^ expected
    assert abs(metrics["cross_entropy"] + metrics["mean_dice_loss"] - metrics["loss"]) < 1e-4
    loss.backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_replayed_step_is_reproducible_at_the_real_shape() -> None:
    """tools/replay_race_full.py: the recorded config-2 step (ViT-Base, 4 views, batch 16) replayed 12 times on identical inputs and masks - every parameter's gradient
    agrees with the first replay's to atomics noise.  With kernels of the real durations on the main, long-axis and two weight-gradient streams, two accumulating
    launches on one buffer running at the same time (round 5: dec_linear, one launch per view) show as a per-cent difference (`--break` re-introduces that bug:
    dec_linear.weight 69 % off, profiles/r05_aa_replay_race_hunt.txt)."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    out = subprocess.run([sys.executable, str(root / "tools" / "replay_race_full.py"), "12"], capture_output=True, text=True, timeout=280, check=False)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("REPLAY RACE HUNT")]
    assert line and line[-1].endswith("clean"), (out.stdout[-600:], out.stderr[-600:])
