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


def test_checkpoint_flow_pretrain_files_to_finetune_and_resume(tmp_path) -> None:  # noqa: ANN001
    """The file flow of the reference's recipes end to end on the GPU, with local files in the released layout:
    pre-train a CineMA for two fused steps -> ``cinema.safetensors`` + ``config.yaml`` -> ``CineMA.from_pretrained`` (same loss as the live model) ->
    ``ConvViT.from_pretrained(config, freeze=False)`` (stem / encoder / fusion arrive, first stem filter tiled over the frames) -> three classification
    steps -> ``save_checkpoint`` / ``load_checkpoint_and_optimizer`` into a fresh model + optimiser (the next step is identical) -> fine-tuned
    ``.safetensors`` + config -> ``ConvViT.from_finetuned`` (same logits)."""
    import yaml
    from safetensors.torch import save_file

    from cinema_amd import CineMA, ConvViT
    from cinema_amd.config import to_config
    from cinema_amd.mae.mae import get_model as get_mae
    from cinema_amd.optim import GradScaler, TrainStep, load_checkpoint_and_optimizer, save_checkpoint

    mae_cfg = {"grad_ckpt": False, "data": {"sax": {"patch_size": [32, 32, 4], "in_chans": 1}, "lax": {"patch_size": [32, 32], "in_chans": 1}},
               "model": {"size": "tiny", "patch_size": [2, 2, 1], "scale_factor": [2, 2, 1], "enc_conv_chans": [8, 16], "enc_conv_n_blocks": 1}}
    torch.manual_seed(0)
    mae = get_mae(to_config(mae_cfg)).to(DEV)
    gen = torch.Generator().manual_seed(5)
    images = {v: torch.rand(2, 1, *(mae_cfg["data"]["sax" if v == "sax" else "lax"]["patch_size"]), generator=gen).to(DEV) for v in mae.views}
    step = TrainStep(mae, lr=1e-3)
    for _ in range(2):
        step(images, 0.75)
    (tmp_path / "config.yaml").write_text(yaml.safe_dump(mae_cfg))
    save_file({k: v.detach().cpu().contiguous() for k, v in mae.state_dict().items()}, str(tmp_path / "cinema.safetensors"))
    loaded = CineMA.from_pretrained(model_path=tmp_path / "cinema.safetensors", config_path=tmp_path / "config.yaml").to(DEV)
    masks = {v: (torch.arange(mae.enc_down_dict[v].patch_embed.n_patches, device=DEV)[None] % 4 != 0).expand(2, -1).contiguous() for v in mae.views}
    with torch.no_grad():
        l0 = float(mae(images, 0.75, enc_mask_dict=masks)[0])
        l1 = float(loaded(images, 0.75, enc_mask_dict=masks)[0])
    assert l0 == pytest.approx(l1, rel=1e-6)  # same weights; the loss reductions use atomics (last-bit order effects)

    # downstream classifier on two of the views, two frames per view
    ft_cfg = {"grad_ckpt": False, "data": {**mae_cfg["data"], "class_column": "label", "label": ["a", "b", "c"]},
              "model": {"name": "convvit", "views": ["sax", "lax_2c"], "n_frames": 2, "out_chans": 3,
                        "convvit": {"size": "tiny", "enc_patch_size": [2, 2, 1], "enc_scale_factor": [2, 2, 1], "enc_conv_chans": [8, 16], "enc_conv_n_blocks": 1,
                                    "drop_path": 0.0}}}
    torch.manual_seed(1)
    clf = ConvViT.from_pretrained(to_config(ft_cfg), freeze=False, model_path=tmp_path / "cinema.safetensors").to(DEV)
    sd, msd = clf.state_dict(), mae.state_dict()
    assert torch.equal(sd["encoder.blocks.0.attn.kv.weight"], msd["encoder.blocks.0.attn.kv.weight"])
    w, wm = sd["enc_down_dict.sax.conv_blocks.0.patch_embed.conv.weight"], msd["enc_down_dict.sax.conv_blocks.0.patch_embed.conv.weight"]
    assert w.shape[1] == 2 and torch.equal(w[:, :1], wm) and torch.equal(w[:, 1:], wm)
    batch = {"sax_image": torch.rand(3, 2, 32, 32, 4, generator=gen), "lax_2c_image": torch.rand(3, 2, 32, 32, generator=gen), "label": torch.tensor([0, 2, 1])}
    ft = ClsTrainStep(clf, VIEWS, lr=1e-3, layer_decay=0.75)
    for _ in range(3):
        ft(batch)
    scaler = GradScaler()
    ck = save_checkpoint(tmp_path / "ckpt", 7, clf, ft.optimizer, scaler, n_samples=9)
    assert ck.name == "ckpt_7.pt"
    torch.manual_seed(2)
    clf2 = get_classification_or_regression_model(to_config(ft_cfg)).to(DEV)
    ft2 = ClsTrainStep(clf2, VIEWS, lr=1e-3, layer_decay=0.75)
    _, _, _, epoch, n_samples = load_checkpoint_and_optimizer(ck, clf2, ft2.optimizer, GradScaler())
    assert (epoch, n_samples) == (7, 9) and ft2.optimizer.step_count == 3
    la, ga, _ = ft(batch)
    lb, gb, _ = ft2(batch)
    assert float(la) == pytest.approx(float(lb), rel=1e-5) and float(ga) == pytest.approx(float(gb), rel=1e-4)
    assert (ft.flat.flat_param - ft2.flat.flat_param).abs().max() <= 1e-5  # one more AdamW step from the restored moments (lr 1e-3)

    (tmp_path / "ft_config.yaml").write_text(yaml.safe_dump(ft_cfg))
    save_file({k: v.detach().cpu().contiguous() for k, v in clf.state_dict().items()}, str(tmp_path / "ft.safetensors"))
    again = ConvViT.from_finetuned(model_path=tmp_path / "ft.safetensors", config_path=tmp_path / "ft_config.yaml").to(DEV)
    clf.eval(), again.eval()
    imgs = {v: batch[f"{v}_image"].to(DEV) for v in VIEWS}
    with torch.no_grad():
        assert (clf(imgs).float() - again(imgs).float()).abs().max() <= 1e-5


def test_train_one_epoch_loop_with_layer_decay_learning_rates() -> None:
    """``cinema_amd.train.train_one_epoch`` (reference ``cinema/train.py:85-168``) around ``ClsTrainStep``: per-iteration LR from the schedule times each group's
    ``lr_scale``, updates on the accumulation boundary, counters and logged keys."""
    import math

    from cinema_amd.config import to_config
    from cinema_amd.train import train_one_epoch

    model, g = _convvit_model()
    cfg = to_config({"train": {"batch_size_per_device": 2, "n_warmup_epochs": 1, "n_epochs": 2, "lr": 1e-3, "min_lr": 1e-6}})
    gen = torch.Generator().manual_seed(3)
    loader = [{**{f"{v}_image": torch.rand(2, *g[f"image/{v}"].shape[1:], generator=gen) for v in VIEWS}, "label": torch.tensor([i % 3, (i + 1) % 3])} for i in range(4)]
    step = ClsTrainStep(model, VIEWS, lr=1e-3, layer_decay=0.75)
    logs, n = [], 0
    for epoch in range(2):
        n = train_one_epoch(step, loader, epoch, 2, n, cfg, log=logs.append)
    assert n == 2 * 4 * 2 and step.optimizer.step_count == 4 and len(logs) == 4 and model.training
    s_last = 3 / 4 + 1
    lr_last = 1e-6 + (1e-3 - 1e-6) * 0.5 * (1 + math.cos(math.pi * (s_last - 1) / (2 - 1)))
    assert logs[-1]["lr"] == pytest.approx(lr_last) and logs[0]["lr"] == pytest.approx(1e-3 * 0.25)
    for grp in step.optimizer.param_groups:
        assert grp["lr"] == pytest.approx(lr_last * grp["lr_scale"])
    assert {"train_cross_entropy", "train_loss", "grad_norm", "lr", "n_samples", "epoch"} <= set(logs[0]) and [lg["epoch"] for lg in logs] == [0, 0, 1, 1]
