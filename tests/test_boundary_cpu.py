"""CPU-side checks of the drop-in boundary: the ``cinema`` import surface, the harness symbols of ``cinema/optim.py`` and
``cinema/device.py``, the rotary module (``cinema/rotary.py`` + its known-answer test ``cinema/rotary_test.py:9-13``)."""

from __future__ import annotations

import math
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp
from torch import nn

from conftest import ROOT, load_golden


# ---------------------------------------------------------------------------------------------------- import surface
def test_cinema_alias_package_serves_the_reference_import_lines() -> None:
    """The import lines of the reference's examples / training scripts (cinema/__init__.py:3-34, examples/train/*.py) resolve to the build."""
    import cinema
    import cinema_amd
    from cinema import CineMA, ConvUNetR, ConvViT, patchify, unpatchify  # noqa: F401
    from cinema.convvit import get_layer_id_for_vit, load_pretrain_weights, param_groups_lr_decay, upsample_mask  # noqa: F401
    from cinema.device import ddp_setup, get_amp_dtype_and_device, get_free_port, print_model_info, setup_ddp_model  # noqa: F401
    from cinema.mae.mae import get_model  # noqa: F401
    from cinema.optim import EarlyStopping, GradScaler, adjust_learning_rate, get_n_accum_steps, load_checkpoint_and_optimizer, save_checkpoint  # noqa: F401
    from cinema.rotary import RotaryEmbedding, apply_rotary_emb, rotate_half  # noqa: F401
    from cinema.segmentation.convunetr import ConvUNetR as SegModel
    from cinema.segmentation.train import segmentation_loss  # noqa: F401
    from cinema.classification.train import classification_forward, classification_loss, get_classification_or_regression_model  # noqa: F401
    from cinema.regression.train import regression_forward, regression_loss  # noqa: F401
    from cinema.vit import get_vit_config  # noqa: F401

    assert CineMA is cinema_amd.CineMA and ConvViT is cinema_amd.ConvViT and SegModel is cinema_amd.ConvUNetR is ConvUNetR
    assert cinema.UKB_SAX_SLICE_SIZE == (192, 192) and cinema.LABEL_TO_NAME == {1: "RV", 2: "MYO", 3: "LV"}
    assert sys.modules["cinema.mae.mae"] is sys.modules["cinema_amd.mae.mae"]


def test_device_helpers() -> None:
    from cinema_amd.device import get_amp_dtype_and_device, print_model_info, setup_ddp_model

    amp, dev = get_amp_dtype_and_device()
    if not torch.cuda.is_available():
        assert amp == torch.float16 and dev.type == "cpu"  # reference cinema/device.py:58-72
    m = nn.Linear(4, 3)
    print_model_info(m)
    a, b = setup_ddp_model(m, torch.device("cpu"), rank=0, world_size=1)
    assert a is m and b is m and not hasattr(m, "grad_synchronizer")


# ---------------------------------------------------------------------------------------------------- rotary (a10)
def test_rotate_half_known_answer() -> None:
    """cinema/rotary_test.py:9-13."""
    from cinema_amd.rotary import rotate_half

    x = torch.tensor([[[[1, 2], [3, 4]]]])
    assert torch.equal(rotate_half(x), torch.tensor([[[[-2, 1], [-4, 3]]]]))


@pytest.mark.parametrize(("n_x_tokens", "n_tokens"), [(8, 8), (8, 10)])
def test_apply_rotary_emb_shapes_and_partial_dim(n_x_tokens: int, n_tokens: int) -> None:
    """cinema/rotary_test.py:16-35, plus the value contract: rotated part is a rotation (norm preserved per pair), the rest passes through."""
    from cinema_amd.rotary import apply_rotary_emb

    torch.manual_seed(0)
    x = torch.rand(2, n_x_tokens, 4, 12)
    ang = torch.rand(n_tokens, 5)
    got = apply_rotary_emb(x, torch.cos(ang), torch.sin(ang))
    assert got.shape == x.shape
    assert torch.equal(got[..., 10:], x[..., 10:])
    n_in = x[..., :5] ** 2 + x[..., 5:10] ** 2
    n_out = got[..., :5] ** 2 + got[..., 5:10] ** 2
    assert torch.allclose(n_in, n_out, atol=1e-5)
    with pytest.raises(ValueError, match="Rotary dim"):
        apply_rotary_emb(torch.rand(2, 8, 4, 8), torch.rand(8, 5), torch.rand(8, 5))


def test_rotary_embedding_module_and_head_tables_vs_reference_table() -> None:
    """cinema/rotary_test.py:38-58 + the table the reference's Attention(rotary=True) really builds (golden 'rotary/cos': rows = heads)."""
    from cinema_amd.rotary import RotaryEmbedding, apply_rotary_emb
    from cinema_amd.vit import Attention

    rot = RotaryEmbedding(12)
    for n in (8, 12):
        q, k = torch.rand(2, n, 4, 12), torch.rand(2, n, 4, 12)
        gq, gk = rot(q, k)
        assert gq.shape == q.shape and gk.shape == k.shape
    with pytest.raises(ValueError, match="same sequence length"):
        rot(torch.rand(2, 8, 4, 12), torch.rand(2, 9, 4, 12))
    assert "inv_freq" not in rot.state_dict()  # non-persistent buffer (rotary.py:82)

    g = load_golden("layers.safetensors")
    attn = Attention(32, n_heads=4, qkv_bias=True, rotary=True)
    assert isinstance(attn.rotary, RotaryEmbedding) and not any("rotary" in k for k in attn.state_dict())
    cos, sin = attn.rotary.head_tables(4, torch.device("cpu"))
    assert torch.allclose(cos, g["rotary/cos"], atol=1e-6)
    # the head-indexed rotation cancels in q.k^T: scores of rotated q, k equal the plain ones
    q, k = torch.rand(2, 4, 6, 8), torch.rand(2, 4, 6, 8)  # (batch, heads, tokens, head_dim) as the reference passes them
    rq, rk = apply_rotary_emb(q, cos, sin), apply_rotary_emb(k, cos, sin)
    assert torch.allclose(rq @ rk.transpose(-1, -2), q @ k.transpose(-1, -2), atol=1e-5)


# ---------------------------------------------------------------------------------------------------- optim harness (a27, f1)
def test_early_stopping_follows_the_reference_rule() -> None:
    """cinema/optim.py:297-330: improvement needs min_delta; has_improved is the plain comparison."""
    from cinema_amd.optim import EarlyStopping

    es = EarlyStopping(min_delta=0.1, patience=2)
    es.update(1.0)
    assert es.best_metric == 1.0 and es.patience_count == 0 and es.has_improved
    es.update(0.95)  # better, but not by min_delta
    assert es.has_improved and es.best_metric == 1.0 and es.patience_count == 1 and not es.should_stop
    es.update(0.8)
    assert es.best_metric == 0.8 and es.patience_count == 0
    es.update(0.9)
    es.update(0.85)
    assert not es.has_improved and es.should_stop


def test_grad_scaler_call_with_a_torch_optimizer_matches_manual_steps(tmp_path: Path) -> None:
    """GradScaler.__call__(loss, optimizer, clip_grad, parameters, update_grad) (cinema/optim.py:183-215) on the torch-optimiser branch:
    accumulation micro-step returns None and leaves the weights, the update step returns the pre-clip norm and applies clip + AdamW;
    save_checkpoint / load_checkpoint_and_optimizer round-trip model, optimiser moments, scaler, epoch and n_samples."""
    from cinema_amd.optim import GradScaler, get_grad_norm, load_checkpoint_and_optimizer, save_checkpoint

    torch.manual_seed(0)
    model, ref = nn.Linear(6, 3), nn.Linear(6, 3)
    ref.load_state_dict(model.state_dict())
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, betas=(0.9, 0.95), weight_decay=0.05)
    opt_ref = torch.optim.AdamW(ref.parameters(), lr=1e-2, betas=(0.9, 0.95), weight_decay=0.05)
    scaler = GradScaler()
    x1, x2 = torch.randn(5, 6), torch.randn(5, 6)
    w0 = model.weight.detach().clone()
    assert scaler(loss=model(x1).pow(2).mean() / 2, optimizer=opt, clip_grad=0.1, parameters=model.parameters(), update_grad=False) is None
    assert torch.equal(model.weight, w0)
    norm = scaler(loss=model(x2).pow(2).mean() / 2, optimizer=opt, clip_grad=0.1, parameters=model.parameters(), update_grad=True)
    opt.zero_grad()
    (ref(x1).pow(2).mean() / 2).backward()
    (ref(x2).pow(2).mean() / 2).backward()
    want = torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.1)
    opt_ref.step()
    assert torch.allclose(norm, want) and torch.allclose(model.weight, ref.weight) and torch.allclose(model.bias, ref.bias)
    with pytest.raises(ValueError, match="parameters must not be None"):
        scaler(loss=model(x1).sum(), optimizer=opt, update_grad=True)
    model(x1).sum().backward()
    assert torch.allclose(get_grad_norm(model.parameters()), torch.sqrt(sum(p.grad.pow(2).sum() for p in model.parameters())))
    opt.zero_grad()

    path = save_checkpoint(tmp_path / "ckpt", epoch=3, model_wo_ddp=model, optimizer=opt, loss_scaler=scaler, n_samples=77)
    assert path.name == "ckpt_3.pt" and set(torch.load(path)) == {"model", "optimizer", "epoch", "scaler", "n_samples"}
    model2 = nn.Linear(6, 3)
    opt2 = torch.optim.AdamW(model2.parameters(), lr=1e-2, betas=(0.9, 0.95), weight_decay=0.05)
    m, o, s, epoch, n_samples = load_checkpoint_and_optimizer(path, model2, opt2, GradScaler())
    assert (epoch, n_samples) == (3, 77) and torch.equal(m.weight, model.weight)
    st, st2 = opt.state_dict()["state"], o.state_dict()["state"]
    assert all(torch.equal(st[i]["exp_avg_sq"], st2[i]["exp_avg_sq"]) for i in st)


def test_flat_model_accepts_layer_decay_param_groups() -> None:
    """FlatModel / FusedAdamW over arbitrary torch-style groups (ConvViT fine-tuning with ``param_groups_lr_decay``, convvit.py:741-810):
    every group is one contiguous range, lr follows lr_scale through adjust_learning_rate."""
    from cinema_amd import CineMA
    from cinema_amd.optim import FlatModel, adjust_learning_rate
    from test_host_cpu import mini_kwargs

    model = CineMA(**mini_kwargs())
    named = dict(model.named_parameters())
    groups = [{"params": [p for n, p in named.items() if n.startswith("encoder.") and p.requires_grad], "weight_decay": 0.05, "lr_scale": 0.5},
              {"params": [p for n, p in named.items() if not n.startswith("encoder.") and p.requires_grad], "weight_decay": 0.0, "lr_scale": 1.0}]
    before = {k: v.clone() for k, v in model.state_dict().items()}
    flat = FlatModel(model, 0.05, param_groups=groups)
    assert len(flat.ranges) == 2 and flat.ranges[0][1] == flat.ranges[1][0] and flat.ranges[1][1] == flat.numel
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k]), k

    class Opt:  # the attribute adjust_learning_rate drives
        param_groups = [{**{k: v for k, v in g.items() if k != "params"}, "lr": 1.0} for g in flat.groups]

    lr = adjust_learning_rate(Opt, step=5, warmup_steps=10, max_n_steps=100, lr=1e-3, min_lr=0.0)
    assert math.isclose(lr, 5e-4) and math.isclose(Opt.param_groups[0]["lr"], 2.5e-4) and math.isclose(Opt.param_groups[1]["lr"], 5e-4)


def test_injected_mask_with_unequal_rows_is_rejected() -> None:
    """The reference fails in its reshape (mae.py:550) when samples mask different counts; here the index lists would be silently wrong."""
    from cinema_amd.convvit import TokenSelection

    mask = torch.zeros(2, 8, dtype=torch.bool)
    mask[0, :6] = True
    mask[1, :5] = True
    with pytest.raises(ValueError, match="same number of patches"):
        TokenSelection(mask, 2, 8, torch.device("cpu"))
    mask[1, 5] = True
    sel = TokenSelection(mask, 2, 8, torch.device("cpu"))
    assert sel.n_drop == 6 and sel.keep.tolist() == [6, 7, 14, 15]


# ---------------------------------------------------------------------------------------------------- collective NaN decision on gloo
def _nan_worker(rank: int, world: int, port: int, tmp: str) -> None:
    sys.path.insert(0, str(ROOT))
    from cinema_amd import CineMA
    from cinema_amd.ddp import GradientSynchronizer, ddp_setup
    from cinema_amd.optim import FlatModel
    from test_host_cpu import mini_kwargs

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    ddp_setup(rank, world, port=port, backend="gloo")
    torch.manual_seed(0)
    flat = FlatModel(CineMA(**mini_kwargs()), 0.05)
    sync = GradientSynchronizer(world, bucket_bytes=64 << 10)
    sync.attach(flat)
    flat.flat_grad.fill_(1.0)
    if rank == 1:
        flat.flat_grad[123] = float("nan")  # a NaN loss on ONE rank back-propagates NaN into that rank's flat gradient buffer
    sync.arm(True)
    sync.all_reduce()
    # what cinema_clip_coef reads on every rank after the exchange: the squared norm of the (mean) gradient
    torch.save(flat.flat_grad.pow(2).sum().sqrt(), f"{tmp}/norm{rank}.pt")
    torch.distributed.destroy_process_group()


def test_nan_on_one_rank_reaches_every_rank_through_the_gradient_exchange(tmp_path: Path) -> None:
    """The skip decision of FusedAdamW.step is taken from the all-reduced gradient norm, so a NaN on one rank makes EVERY rank skip
    (no rank-local `continue` as in cinema/mae/pretrain.py:255-257, which would leave the other ranks waiting in a collective)."""
    from cinema_amd.ddp import get_free_port

    mp.spawn(_nan_worker, args=(2, get_free_port(), str(tmp_path)), nprocs=2, join=True)
    n0, n1 = torch.load(tmp_path / "norm0.pt"), torch.load(tmp_path / "norm1.pt")
    assert not torch.isfinite(n0) and not torch.isfinite(n1)


def test_metric_helpers_follow_the_reference_formulas() -> None:
    """``cinema.metric`` scalars and host-tensor volumes (reference ``cinema/metric.py:84-146``)."""
    import numpy as np
    from cinema.metric import NORMAL_EF, REDUCED_EF, coefficient_of_variance, ejection_fraction, get_ef_region, get_volumes

    assert (REDUCED_EF, NORMAL_EF) == (40, 55)
    assert ejection_fraction(120.0, 50.0) == pytest.approx(58.3333333)
    assert np.allclose(ejection_fraction(np.array([100.0, 80.0]), np.array([40.0, 60.0])), [60.0, 25.0])
    assert [get_ef_region(v) for v in (10.0, 40.0, 40.1, 55.0, 55.1)] == [0, 0, 1, 1, 2]
    x, y = np.array([1.0, 2.0, 4.0]), np.array([1.1, 1.9, 4.4])
    assert coefficient_of_variance(x, y) == pytest.approx(float(np.sqrt(np.mean(((x - y) ** 2 / 2) / ((x + y) / 2) ** 2))))
    mask = torch.zeros(2, 3, 4, 5, 2)
    mask[0, 1, :2] = 1
    mask[1, 2, :, :3] = 1
    vol = get_volumes(mask, (1.5, 1.5, 8.0))
    assert vol.shape == (2, 3) and vol[0, 1] == pytest.approx(2 * 5 * 2 * 18.0 / 1000) and vol[1, 2] == pytest.approx(4 * 3 * 2 * 18.0 / 1000) and float(vol[0, 0]) == 0.0


def test_module_level_helper_symbols_of_the_reference_resolve() -> None:
    """The free functions the reference exports beside its classes (cinema/mae/mae.py:68-152, cinema/vit.py:67-225,386-405, cinema/optim.py:55-119) import from the
    alias package; the pure host ones agree with the reference's definitions here (their device twins are checked in tests/test_boundary_gpu.py)."""
    import numpy as np
    import torch

    from cinema.mae.mae import add_pos_embed_and_append_mask_token, mse_loss  # noqa: F401
    from cinema.optim import CosineScheduler, apply_optim_scheduler
    from cinema.vit import (get_nd_sincos_pos_embed_from_grid, patchify, patchify_2d, patchify_3d, patchify_4d, unpatchify_2d, unpatchify_3d,  # noqa: F401
                            unpatchify_4d)

    x2, x3, x4 = torch.randn(2, 3, 8, 12), torch.randn(2, 1, 8, 6, 4), torch.randn(1, 2, 4, 4, 2, 6)
    assert torch.equal(patchify_2d(x2, (4, 4)), patchify(x2, (4, 4))) and torch.equal(unpatchify_2d(patchify_2d(x2, (4, 4)), (4, 4), (2, 3)), x2)
    assert torch.equal(unpatchify_3d(patchify_3d(x3, (4, 2, 1)), (4, 2, 1), (2, 3, 4)), x3)
    assert torch.equal(unpatchify_4d(patchify_4d(x4, (2, 2, 1, 3)), (2, 2, 1, 3), (2, 2, 2, 2)), x4)
    with pytest.raises(ValueError, match="cannot be divided"):
        patchify_2d(x2, (3, 4))
    # the reference formula: even width per axis, zero padding of the rest (cinema/vit.py:398-405)
    grid = np.stack(np.meshgrid(np.arange(3, dtype=np.float32), np.arange(2, dtype=np.float32), np.arange(4, dtype=np.float32)), axis=0)
    emb = get_nd_sincos_pos_embed_from_grid(16, grid)
    assert emb.shape == (24, 16) and np.all(emb[:, 12:] == 0) and np.allclose(emb[0, :4], [0, 0, 1, 1])
    # CosineScheduler known answers: freeze zeros, linear warm-up, half cosine, final value past the end
    s = CosineScheduler(1.0, 0.1, 100, warmup_iters=10, freeze_iters=5)
    assert s[0] == 0.0 and s[4] == 0.0 and s[5] == 0.0 and abs(s[14] - 1.0) < 1e-12 and abs(s[15] - 1.0) < 1e-12 and s[100] == 0.1 and s[1000] == 0.1
    assert abs(s[15 + 42] - (0.1 + 0.45 * (1 + math.cos(math.pi * 42 / 85)))) < 1e-12
    with pytest.raises(ValueError, match="Length of schedule"):
        CosineScheduler(1.0, 0.1, 10, warmup_iters=8, freeze_iters=5)
    lin = torch.nn.Linear(2, 2)
    opt = torch.optim.SGD([{"params": [lin.weight], "lr_scale": 0.5, "weight_decay_scale": 1.0, "is_last_layer": False},
                           {"params": [lin.bias], "lr_scale": 1.0, "weight_decay_scale": 0.0, "is_last_layer": True}], lr=1.0)
    apply_optim_scheduler(opt, lr=0.2, last_layer_lr=0.05, weight_decay=0.1)
    assert [g["lr"] for g in opt.param_groups] == [0.1, 0.05] and [g["weight_decay"] for g in opt.param_groups] == [0.1, 0.0]


def test_block_options_are_accepted_with_the_reference_parameter_names() -> None:
    """``init_values`` (timm LayerScale), ``qk_norm`` and ``proj_drop`` of ``cinema/vit.py:446-609`` and ``PatchEmbed(dynamic_img_pad=True)`` construct (rounds 1-5
    raised NotImplementedError) and their parameters carry the reference's names; ``attn_drop`` still has no HIP path and says so."""
    import pytest
    from torch import nn

    from cinema.vit import Attention, Block, PatchEmbed
    from cinema_amd.vit import Mlp

    blk = Block(dim=32, n_heads=4, mlp_ratio=2, norm_layer=nn.LayerNorm, norm_eps=1e-6, drop_path=0.0, qkv_bias=True, rotary=False, act_layer=nn.GELU, mlp_layer=Mlp,
                qk_norm=True, proj_drop=0.1, init_values=1e-5)
    keys = set(blk.state_dict())
    assert {"ls1.gamma", "ls2.gamma", "attn.q_norm.weight", "attn.q_norm.bias", "attn.k_norm.weight", "attn.k_norm.bias"} <= keys
    assert float(blk.ls1.gamma[0]) == pytest.approx(1e-5) and blk.attn.q_norm.normalized_shape == (8,)
    assert blk.attn.proj_drop.p == 0.1 and blk.mlp.drop1.p == 0.1 and blk.mlp.drop2.p == 0.1
    assert PatchEmbed(image_size=(30, 30), patch_size=(4, 4), in_chans=1, embed_dim=8, dynamic_img_pad=True).dynamic_img_pad
    with pytest.raises(NotImplementedError):
        Attention(32, n_heads=4, attn_drop=0.1)
