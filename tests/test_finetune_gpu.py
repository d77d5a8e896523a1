"""GPU checks of the ConvViT fine-tuning path: the head-loss kernels, ``classification_loss`` / ``regression_loss`` and the
patch-averaged evaluation forwards against the reference's golden vectors (tests/golden/convvit_heads.safetensors, oracle/make_golden_heads.py),
and the fused fine-tuning step with layer-decay parameter groups.

Tolerances: the head-loss kernels compute in fp32 (1e-5 against the oracle); through the bf16 model the loss is held to 2e-2 absolute (logits max-abs
3e-2 as in the ConvViT golden test), gradients to relative L2 6e-2 (the bound of the other model-gradient tests)."""

from __future__ import annotations

import types

import pytest
import torch

pytestmark = pytest.mark.gpu

import cinema_oracle as O  # noqa: E402
from cinema_amd import hip as K  # noqa: E402
from cinema_amd.classification.train import (ClsTrainStep, classification_forward, classification_loss, classification_loss_tensors, cross_entropy,  # noqa: E402
                                             get_classification_or_regression_model)
from cinema_amd.regression.train import RegTrainStep, regression_forward, regression_loss  # noqa: E402
from conftest import load_golden  # noqa: E402
from test_model_gpu import _convvit_model, split  # noqa: E402

DEV = "cuda"
VIEWS = ["sax", "lax_2c"]


def test_head_ce_kernel_matches_oracle_and_autograd() -> None:
    g = torch.Generator().manual_seed(0)
    for b, c, eps in ((5, 7, 0.1), (2, 3, 0.0), (300, 2, 0.2)):  # more rows than the block has threads in the last case
        logits = torch.randn(b, c, generator=g) * 3
        labels = torch.randint(0, c, (b,), generator=g)
        ref_in = logits.clone().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(ref_in, labels, label_smoothing=eps)
        ref.backward()
        assert abs(float(O.classification_loss_value(logits, labels, eps)) - float(ref)) <= 1e-5
        out, d = K.head_ce(logits.to(DEV), labels.to(torch.int32).to(DEV), eps)
        assert abs(float(out) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
        assert (d.cpu() - ref_in.grad).abs().max() <= 1e-6
    x = torch.randn(4, 3, device=DEV, requires_grad=True)
    (cross_entropy(x, torch.tensor([0, 1, 2, 1], device=DEV), 0.1) * 2.0).backward()  # upstream factor reaches the gradient
    ref_in = x.detach().cpu().requires_grad_(True)
    (torch.nn.functional.cross_entropy(ref_in, torch.tensor([0, 1, 2, 1]), label_smoothing=0.1) * 2.0).backward()
    assert (x.grad.cpu() - ref_in.grad).abs().max() <= 1e-6
    with pytest.raises(ValueError):
        cross_entropy(x, torch.tensor([0, 1], device=DEV))
    with pytest.raises(K.HipLibraryError):
        K.head_ce(x.detach(), torch.zeros(4, dtype=torch.int32, device=DEV), 1.0)


def test_head_mse_kernel_matches_oracle() -> None:
    g = torch.Generator().manual_seed(1)
    for shape in ((2, 3), (7, 1), (129, 5)):
        pred, label = torch.randn(*shape, generator=g), torch.randn(*shape, generator=g) * 2
        out, d = K.head_mse(pred.to(DEV), label.to(DEV))
        vals = O.regression_loss_values(pred, label)
        for i, k in enumerate(("mse_loss", "mae_loss", "max_label", "min_label", "max_pred", "min_pred")):
            assert abs(float(out[i]) - vals[k]) <= 1e-5 * max(1.0, abs(vals[k])), (shape, k)
        assert (d.cpu() - 2 * (pred - label) / pred.numel()).abs().max() <= 1e-6


def test_classification_and_regression_loss_vs_reference_golden() -> None:
    model, g = _convvit_model()
    h = load_golden("convvit_heads.safetensors")
    named = dict(model.named_parameters())
    batch = {f"{v}_image": g[f"image/{v}"] for v in VIEWS}
    batch["label"] = h["cls/label"]
    loss, metrics = classification_loss(model, batch, VIEWS, torch.device(DEV), label_smoothing=0.1)
    assert set(metrics) == {"cross_entropy", "loss"} and all(isinstance(v, float) for v in metrics.values())
    assert abs(float(loss) - float(h["cls/loss"])) <= 2e-2 and abs(metrics["cross_entropy"] - float(h["cls/loss"])) <= 2e-2
    loss.backward()
    for k, t in split(h, "cls/grad/").items():
        rel = float((named[k].grad.float().cpu() - t).norm() / (t.norm() + 1e-12))
        assert rel <= 6e-2, (k, rel)
    model.zero_grad()
    batch["label"] = h["reg/label"]
    loss, metrics = regression_loss(model, batch, VIEWS, torch.device(DEV))
    assert set(metrics) == {"mse_loss", "loss", "mae_loss", "max_label", "min_label", "max_pred", "min_pred"}
    for i, k in enumerate(("mse_loss", "mae_loss", "max_label", "min_label", "max_pred", "min_pred", "loss")):
        assert abs(metrics[k] - float(h["reg/metrics"][i])) <= 3e-2, (k, metrics[k], float(h["reg/metrics"][i]))
    loss.backward()
    for k, t in split(h, "reg/grad/").items():
        rel = float((named[k].grad.float().cpu() - t).norm() / (t.norm() + 1e-12))
        assert rel <= 6e-2, (k, rel)
    with pytest.raises(ValueError):
        regression_loss(model, {**batch, "label": torch.zeros(2, 2)}, VIEWS, torch.device(DEV))


def test_patch_averaged_forward_vs_reference_golden() -> None:
    model, _ = _convvit_model()
    model.eval()
    h = load_golden("convvit_heads.safetensors")
    images = {k: v.to(DEV) for k, v in split(h, "fwd/image/").items()}
    sizes = {"sax": (32, 32, 4), "lax_2c": (32, 32)}
    out = classification_forward(model, images, sizes, torch.bfloat16)
    assert out.shape == (1, 3) and (out.float().cpu() - h["fwd/cls_logits"]).abs().max() <= 3e-2
    assert abs(float(torch.exp(out.float()).sum()) - 1.0) <= 1e-4  # log of a probability vector
    out = regression_forward(model, images, sizes, torch.bfloat16)
    assert out.shape == (1, 3) and (out.float().cpu() - h["fwd/reg_preds"]).abs().max() <= 3e-2
    whole = {"sax": images["sax"][:, :, :32, :32].contiguous(), "lax_2c": images["lax_2c"]}
    out = classification_forward(model, whole, sizes, torch.bfloat16)
    assert (out.float().cpu() - h["fwd/cls_logits_whole"]).abs().max() <= 3e-2
    with pytest.raises(ValueError):  # two over-sized views
        classification_forward(model, {"sax": images["sax"], "lax_2c": torch.rand(1, 2, 40, 32, device=DEV)}, sizes)
    with pytest.raises(ValueError):  # smaller than the patch
        regression_forward(model, {"sax": images["sax"][:, :, :16], "lax_2c": images["lax_2c"]}, sizes)


@pytest.mark.parametrize("task", ["classification", "regression"])
def test_fused_finetune_step_learns_a_fixed_batch(task: str) -> None:
    """``ClsTrainStep`` / ``RegTrainStep`` (layer-decay groups of ``param_groups_lr_decay``, flat-buffer clip + fused AdamW): the loss of a fixed batch falls,
    the gradient norm is finite, the step counter advances and the metrics stay on the device."""
    model, g = _convvit_model()
    model.train()
    gen = torch.Generator().manual_seed(7)
    cls = torch.tensor([0, 1, 2, 1])
    # learnable: the mean intensity of a sample is a function of its label
    batch = {f"{v}_image": 0.3 * torch.rand(4, *g[f"image/{v}"].shape[1:], generator=gen) + 0.3 * cls.float().reshape(4, *[1] * (g[f"image/{v}"].dim() - 1)) for v in VIEWS}
    if task == "classification":
        batch["label"] = cls
        step = ClsTrainStep(model, VIEWS, label_smoothing=0.1, lr=1e-2, layer_decay=0.75)
    else:
        batch["label"] = torch.stack([cls.float() - 1.0, 0.5 * cls.float(), torch.ones(4)], dim=1)
        step = RegTrainStep(model, VIEWS, lr=1e-2, layer_decay=0.75)
    assert len({gr["lr_scale"] for gr in step.flat.groups}) > 2  # the layer-decay scales reached the optimiser
    losses = []
    for _ in range(30):
        loss, gnorm, metrics = step(batch)
        assert torch.isfinite(gnorm) and all(torch.is_tensor(v) and v.is_cuda for v in metrics.values())
        losses.append(float(loss))
    assert step.optimizer.step_count == 30 and step.optimizer.n_skipped == 0
    assert losses[-1] < 0.8 * losses[0], losses
    # gradient accumulation: two half-weighted calls on the same batch == one call (same gradient, so the same update up to rounding)
    loss_a, gn_a, _ = step(batch, n_accum_steps=2, update_grad=False)
    assert gn_a is None
    _, gn_b, _ = step(batch, n_accum_steps=2, update_grad=True)
    assert torch.isfinite(gn_b)


def test_model_builder_and_tensor_metrics() -> None:
    cfg = types.SimpleNamespace(model=types.SimpleNamespace(name="resnet"))
    with pytest.raises(ValueError):
        get_classification_or_regression_model(cfg)
    model, g = _convvit_model()
    batch = {f"{v}_image": g[f"image/{v}"] for v in VIEWS}
    batch["label"] = torch.tensor([1, 1])
    loss, metrics = classification_loss_tensors(model, batch, VIEWS, torch.device(DEV))
    assert loss.requires_grad and not metrics["loss"].requires_grad and float(metrics["loss"]) == float(loss)
