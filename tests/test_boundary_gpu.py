"""GPU checks of the boundary pieces added around the MAE step: the head-indexed rotary kernel and ``Attention(rotary=True)``, the
device-side non-finite guard of the fused optimiser, checkpoint save / resume of the flat-buffer optimiser through the reference-shaped
``save_checkpoint`` / ``load_checkpoint_and_optimizer``, the ``GradScaler``-shaped call on the fused optimiser, recorded steps fed with
non-fp32 / non-contiguous inputs, and layer-decay parameter groups on the fused optimiser."""

from __future__ import annotations

import math
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu

import cinema_oracle as O  # noqa: E402
from cinema_amd import CineMA  # noqa: E402
from cinema_amd import hip as K  # noqa: E402
from conftest import load_golden  # noqa: E402
from test_model_gpu import mini_kwargs, model_sizes, split  # noqa: E402

DEV = "cuda"


# ---------------------------------------------------------------------------------------------------- rotary (a10)
@pytest.mark.parametrize(("heads", "hd", "ro"), [(4, 8, 8), (12, 64, 64), (16, 32, 32), (4, 16, 8)])
def test_rope_heads_kernel_vs_reference_formula(heads: int, hd: int, ro: int) -> None:
    """cinema_rope_heads on fused q|k rows == apply_rotary_emb as the reference calls it (q, k as (batch, heads, tokens, head_dim), table rows =
    heads; cinema/vit.py:496-499, cinema/rotary.py:27-60), incl. a partial rotary dim; inverse=1 undoes it (orthogonal rotation)."""
    from cinema_amd.rotary import apply_rotary_emb

    torch.manual_seed(0)
    b, t = 2, 37
    c = heads * hd
    x = torch.randn(b * t, 3 * c).bfloat16()
    ang = torch.rand(heads, ro // 2)
    cos, sin = torch.cos(ang), torch.sin(ang)
    xf = x.float()
    want = xf.clone()
    for part in range(2):  # q and k thirds of the fused projection
        blk = xf[:, part * c:(part + 1) * c].reshape(b, t, heads, hd).permute(0, 2, 1, 3)  # (b, heads, t, hd)
        rot = apply_rotary_emb(blk, cos, sin).permute(0, 2, 1, 3).reshape(b * t, c)
        want[:, part * c:(part + 1) * c] = rot
    xg = x.to(DEV)
    K.rope_heads(xg, 2 * heads, heads, hd, cos.to(DEV), sin.to(DEV))
    got = xg.float().cpu()
    assert torch.equal(got[:, 2 * c:], xf[:, 2 * c:])  # v untouched
    assert (got - want).abs().max() <= 2.0 ** -7 * want.abs().max()  # one bf16 rounding of the result
    K.rope_heads(xg, 2 * heads, heads, hd, cos.to(DEV), sin.to(DEV), inverse=True)
    assert (xg.float().cpu() - xf).abs().max() <= 2.0 ** -6 * xf.abs().max()


def test_attention_with_rotary_vs_reference_golden() -> None:
    """Attention(rotary=True) on the HIP path against the reference layer output (tests/golden/layers.safetensors: rotary/out), forward and
    input gradient against autograd through the oracle's attention on the same weights."""
    from cinema_amd.vit import Attention

    g = load_golden("layers.safetensors")
    attn = Attention(32, n_heads=4, qkv_bias=True, rotary=True)
    attn.load_state_dict(split(g, "rotary/param/"))
    attn.to(DEV)
    x = g["rotary/x"].to(DEV).requires_grad_(True)
    y = attn(x)
    err = (y.float().cpu() - g["rotary/out"]).abs().max()
    assert err <= 2e-2 * g["rotary/out"].abs().max() + 2e-3, float(err)
    y.square().sum().backward()
    p = {f"a.{k}": v.clone().requires_grad_(True) for k, v in split(g, "rotary/param/").items()}
    xr = g["rotary/x"].clone().requires_grad_(True)
    O.attention(xr, xr, p, "a", 4).square().sum().backward()
    assert (x.grad.cpu() - xr.grad).norm() <= 5e-2 * xr.grad.norm()
    got_w = attn.kv.weight.grad.cpu()
    assert (got_w - p["a.kv.weight"].grad).norm() <= 5e-2 * p["a.kv.weight"].grad.norm()


def test_cinema_model_with_rotary_matches_plain_model() -> None:
    """rotary=True through the whole MAE step: the head-indexed rotation cancels in q.k^T, so loss and gradients equal the rotary=False model's
    up to the extra bf16 rounding of q and k (the reference: 'rotary/out' == 'rotary/out_plain' to 6e-8 in fp32)."""
    torch.manual_seed(0)
    plain = CineMA(**mini_kwargs(cross_attn=False))  # the reference refuses rotary with separate keys (cinema/vit.py:494-495): self-attention decoder
    rot = CineMA(**mini_kwargs(cross_attn=False, rotary=True))
    with pytest.raises(ValueError, match="not supported with different query and key"):
        bad = CineMA(**mini_kwargs(rotary=True)).to(DEV)
        bad({v: torch.rand(1, 1, *s, device=DEV) for v, s in model_sizes(bad).items()}, 0.75)
    rot.load_state_dict(plain.state_dict())
    assert rot.encoder.blocks[0].attn.rotary is not None and plain.encoder.blocks[0].attn.rotary is None
    plain.to(DEV)
    rot.to(DEV)
    gen = torch.Generator().manual_seed(3)
    images = {v: torch.rand(2, 1, *s, generator=gen).to(DEV) for v, s in model_sizes(plain).items()}
    masks = {v: O.random_patch_mask(2, math.prod(plain.enc_down_dict[v].patch_embed.grid_size), 0.75, gen).to(DEV) for v in images}
    out = {}
    for name, m in (("plain", plain), ("rot", rot)):
        loss, _, _, _ = m(images, 0.75, enc_mask_dict=masks)
        loss.backward()
        out[name] = (float(loss), m.encoder.blocks[0].attn.kv.weight.grad.float().cpu())
    assert abs(out["plain"][0] - out["rot"][0]) <= 5e-3 * abs(out["plain"][0])
    assert (out["plain"][1] - out["rot"][1]).norm() <= 5e-2 * out["plain"][1].norm()


# ---------------------------------------------------------------------------------------------------- non-finite guard (a27)
def test_clip_coef_and_adamw_skip_on_a_non_finite_norm() -> None:
    """cinema_clip_coef writes coef = 0 and counts a skipped update when the squared norm is NaN / inf; cinema_adamw then leaves parameters,
    moments and the bf16 shadow untouched; a finite norm advances the device step count that feeds the bias corrections."""
    n = 1000
    torch.manual_seed(0)
    p, g = torch.randn(n, device=DEV), torch.randn(n, device=DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    shadow = p.bfloat16()
    coef, norm = torch.ones(1, device=DEV), torch.zeros(1, device=DEV)
    state = torch.zeros(2, dtype=torch.int32, device=DEV)
    for bad in (float("nan"), float("inf")):
        sq = torch.tensor([bad], device=DEV)
        K.clip_coef(sq, 5.0, coef, norm, state)
        p0, s0 = p.clone(), shadow.clone()
        K.adamw(p, g, m, v, 1e-3, 0.9, 0.95, 1e-8, 0.05, 1, clip=coef, shadow=shadow, step_state=state)
        assert float(coef) == 0.0 and torch.equal(p, p0) and torch.equal(shadow, s0) and float(m.abs().sum()) == 0.0 and float(v.abs().sum()) == 0.0
    assert state.tolist() == [0, 2]
    ref_p = p.detach().cpu().clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref_p], lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    for _ in range(3):
        sq = (g * g).sum().reshape(1)
        K.clip_coef(sq, 5.0, coef, norm, state)
        K.adamw(p, g, m, v, 1e-3, 0.9, 0.95, 1e-8, 0.05, 12345, clip=coef, shadow=shadow, step_state=state)  # the host step argument is ignored
        ref_p.grad = g.cpu().clone()
        torch.nn.utils.clip_grad_norm_([ref_p], 5.0)
        opt.step()
    assert state.tolist() == [3, 2]
    assert torch.allclose(p.cpu(), ref_p.detach(), rtol=1e-5, atol=1e-6)
    assert torch.equal(shadow, p.bfloat16())


def test_train_step_skips_a_nan_batch_and_recovers() -> None:
    """A NaN input gives a NaN loss (a value, not an exception: cinema/mae/mae.py:604-608 keeps finite views only, all views NaN here); the
    optimisation step must not touch the model (reference: pretrain.py:255-257 `continue`, GradScaler.step inf/NaN skip) and the next clean
    batch must train normally."""
    from cinema_amd.optim import TrainStep

    torch.manual_seed(0)
    model = CineMA(**mini_kwargs()).to(DEV)
    step = TrainStep(model, lr=1e-3)
    gen = torch.Generator().manual_seed(1)
    good = {v: torch.rand(2, 1, *s, generator=gen).to(DEV) for v, s in model_sizes(model).items()}
    l0, g0, _ = step(good, 0.75)
    assert math.isfinite(float(l0)) and step.optimizer.step_count == 1
    before = (step.flat.flat_param.clone(), step.optimizer.exp_avg.clone(), step.optimizer.exp_avg_sq.clone(), step.flat.flat_shadow.clone())
    bad = {k: v.clone() for k, v in good.items()}
    for v in bad.values():
        v.view(-1)[::7] = float("nan")
    l1, g1, _ = step(bad, 0.75)
    assert math.isnan(float(l1)) and not math.isfinite(float(g1))
    after = (step.flat.flat_param, step.optimizer.exp_avg, step.optimizer.exp_avg_sq, step.flat.flat_shadow)
    assert all(torch.equal(a, b) for a, b in zip(before, after))
    assert step.optimizer.step_count == 1 and step.optimizer.n_skipped == 1 and float(step.flat.flat_grad.abs().sum()) == 0.0
    l2, g2, _ = step(good, 0.75)
    assert math.isfinite(float(l2)) and math.isfinite(float(g2)) and float(l2) < float(l0) and step.optimizer.step_count == 2
    assert not torch.equal(before[0], step.flat.flat_param)


# ---------------------------------------------------------------------------------------------------- checkpoint / GradScaler call (f1, b)
def test_fused_optimizer_checkpoint_round_trip_and_grad_scaler_call(tmp_path: Path) -> None:
    """save_checkpoint / load_checkpoint_and_optimizer (cinema/optim.py:229-294) with the flat-buffer optimiser: masters, both Adam moments,
    the step count and the learning rates come back bit-identical in a FRESH model + optimiser, the bf16 shadows are re-derived, and the
    resumed run continues the original trajectory.  The steps are driven through the reference-shaped GradScaler call."""
    from cinema_amd.optim import FlatModel, FusedAdamW, GradScaler, adjust_learning_rate, load_checkpoint_and_optimizer, save_checkpoint

    def make():  # noqa: ANN202
        torch.manual_seed(0)
        model = CineMA(**mini_kwargs()).to(DEV)
        return model, FusedAdamW(FlatModel(model, 0.05), lr=1e-3)

    gen = torch.Generator().manual_seed(5)
    probe = CineMA(**mini_kwargs())
    sizes = model_sizes(probe)
    n_patches = {v: probe.enc_down_dict[v].patch_embed.n_patches for v in sizes}
    batches = [{v: torch.rand(2, 1, *s, generator=gen).to(DEV) for v, s in sizes.items()} for _ in range(4)]
    masks = [{v: O.random_patch_mask(2, n_patches[v], 0.75, gen).to(DEV) for v in sizes} for _ in range(4)]

    def run(model, opt, scaler, i):  # noqa: ANN001, ANN202
        adjust_learning_rate(opt, i / 4, 1, 5, 1e-3, 1e-6)
        loss, _, _, _ = model(batches[i], 0.75, enc_mask_dict=masks[i])
        norm = scaler(loss=loss, optimizer=opt, clip_grad=5.0, parameters=model.parameters(), update_grad=True)
        opt.zero_grad()
        return float(loss), float(norm)

    model, opt = make()
    scaler = GradScaler()
    for i in range(2):
        run(model, opt, scaler, i)
    path = save_checkpoint(tmp_path, epoch=1, model_wo_ddp=model, optimizer=opt, loss_scaler=scaler, n_samples=4)
    tail = [run(model, opt, scaler, i) for i in (2, 3)]

    model2, opt2 = make()
    run(model2, opt2, GradScaler(), 3)  # dirty the fresh state first: everything must come from the file
    _, _, _, epoch, n_samples = load_checkpoint_and_optimizer(path, model2, opt2, GradScaler())
    assert (epoch, n_samples) == (1, 4) and opt2.step_count == 2
    ck = torch.load(path, map_location="cpu")
    assert torch.equal(opt2.exp_avg.cpu(), ck["optimizer"]["exp_avg"].cpu()) and torch.equal(opt2.exp_avg_sq.cpu(), ck["optimizer"]["exp_avg_sq"].cpu())
    for k, v in model2.state_dict().items():
        assert torch.equal(v.cpu(), ck["model"][k].cpu()), k
    assert torch.equal(opt2.flat.flat_shadow, opt2.flat.flat_param.bfloat16())
    tail2 = [run(model2, opt2, GradScaler(), i) for i in (2, 3)]
    for (la, na), (lb, nb) in zip(tail, tail2):
        assert abs(la - lb) <= 2e-4 * abs(la) and abs(na - nb) <= 2e-3 * abs(na), (tail, tail2)


def test_recorded_step_with_bf16_and_non_contiguous_inputs() -> None:
    """The static input tensors of a recording are fp32 contiguous whatever the caller passes (the forward's .float().contiguous() would
    otherwise be an ATen copy outside the launch list and every replay would train on the first batch): replays on bf16 / permuted
    batches must give the eager loss of those batches."""
    from cinema_amd.optim import TrainStep

    sizes = {"sax": (32, 32, 4)}
    kw = mini_kwargs()
    for key in ("image_size_dict", "in_chans_dict", "enc_patch_size_dict", "enc_scale_factor_dict"):
        kw[key] = {"sax": kw[key]["sax"]}
    gen = torch.Generator().manual_seed(9)
    raw = [torch.rand(2, 1, 4, 32, 32, generator=gen) for _ in range(3)]
    feeds = [[r.permute(0, 1, 3, 4, 2).to(DEV) for r in raw],                      # non-contiguous fp32 views
             [r.permute(0, 1, 3, 4, 2).contiguous().bfloat16().to(DEV) for r in raw]]  # bf16
    for batches in feeds:
        losses = {}
        for mode in ("eager", "replay"):
            torch.manual_seed(7)
            model = CineMA(**kw).to(DEV)
            step = TrainStep(model, lr=0.0, replay=(mode == "replay"))  # lr 0: the loss of a step depends on its batch only
            torch.manual_seed(21)
            losses[mode] = [float(step({"sax": b}, 0.75)[0]) for b in batches]
        assert len(set(round(x, 6) for x in losses["eager"])) == 3, losses  # three different batches
        for a, b in zip(losses["eager"], losses["replay"]):
            assert abs(a - b) <= 2e-4 * abs(a), losses
    assert sizes["sax"] == tuple(feeds[0][0].shape[2:])


def test_train_step_with_layer_decay_groups_matches_per_group_adamw() -> None:
    """TrainStep(param_groups=...) (ConvViT fine-tuning, convvit.py:741-810 groups with lr_scale): one fused AdamW launch per group range with
    the group's lr and weight decay == torch.optim.AdamW over the same groups on the same gradients."""
    from cinema_amd.optim import FlatModel, FusedAdamW, adjust_learning_rate

    torch.manual_seed(0)
    model = CineMA(**mini_kwargs()).to(DEV)
    named = dict(model.named_parameters())
    enc = [p for n, p in named.items() if n.startswith("encoder.") and p.requires_grad]
    rest = [p for n, p in named.items() if not n.startswith("encoder.") and p.requires_grad]
    groups = [{"params": enc, "weight_decay": 0.05, "lr_scale": 0.25}, {"params": rest, "weight_decay": 0.0, "lr_scale": 1.0}]
    ref = {n: p.detach().cpu().clone().requires_grad_(p.requires_grad) for n, p in named.items()}
    ref_groups = [{"params": [ref[n] for n, p in named.items() if n.startswith("encoder.") and p.requires_grad], "weight_decay": 0.05, "lr_scale": 0.25},
                  {"params": [ref[n] for n, p in named.items() if not n.startswith("encoder.") and p.requires_grad], "weight_decay": 0.0, "lr_scale": 1.0}]
    opt = FusedAdamW(FlatModel(model, 0.05, param_groups=groups), lr=1e-3)
    ropt = torch.optim.AdamW(ref_groups, lr=1e-3, betas=(0.9, 0.95))
    gen = torch.Generator().manual_seed(2)
    for i in range(3):
        adjust_learning_rate(opt, i + 1, 2, 10, 1e-3, 1e-5)
        adjust_learning_rate(ropt, i + 1, 2, 10, 1e-3, 1e-5)
        assert [g["lr"] for g in opt.param_groups] == [g["lr"] for g in ropt.param_groups]
        for n, p in named.items():
            if p.requires_grad:
                gval = torch.randn(p.shape, generator=gen) * 0.01
                p.grad.copy_(gval.to(DEV))
                ref[n].grad = gval.clone()
        norm = opt.step(clip_grad=None)
        ropt.step()
        want = torch.sqrt(sum(r.grad.pow(2).sum() for r in ref.values() if r.grad is not None))
        assert abs(float(norm) - float(want)) <= 1e-4 * float(want)
    for n, p in named.items():
        if p.requires_grad:
            assert torch.allclose(p.detach().cpu(), ref[n].detach(), rtol=2e-5, atol=2e-7), n


def test_pretrain_one_epoch_loop_follows_the_reference_schedule() -> None:
    """``cinema_amd.mae.pretrain.pretrain_one_epoch`` (reference ``pretrain.py:203-284``): learning rate of every iteration = ``adjust_learning_rate`` at the
    fractional epoch, optimiser updates only on the accumulation boundary, sample counter, logged keys; and the loss of a repeated synthetic set falls."""
    import math

    from torch.utils.data import DataLoader

    from cinema_amd.config import to_config
    from cinema_amd.mae.mae import get_model
    from cinema_amd.mae.pretrain import SyntheticCine, pretrain_one_epoch
    from cinema_amd.optim import TrainStep, get_n_accum_steps

    cfg = to_config({"grad_ckpt": False, "data": {"sax": {"patch_size": [32, 32, 4], "in_chans": 1}, "lax": {"patch_size": [32, 32], "in_chans": 1}},
                     "model": {"size": "tiny", "patch_size": [2, 2, 1], "scale_factor": [2, 2, 1], "enc_conv_chans": [8, 16], "enc_conv_n_blocks": 1},
                     "train": {"batch_size_per_device": 2, "batch_size": 4, "enc_mask_ratio": 0.75, "n_warmup_epochs": 1, "n_epochs": 3, "lr": 2e-3,
                               "min_lr": 1e-5}})
    torch.manual_seed(0)
    model = get_model(cfg).to(DEV)
    ds = SyntheticCine({"sax": (32, 32, 4), "lax_2c": (32, 32), "lax_3c": (32, 32), "lax_4c": (32, 32)}, dict.fromkeys(model.views, 1), length=8, seed=1)
    loader = DataLoader(ds, batch_size=2, shuffle=False)
    n_accum = get_n_accum_steps(batch_size=cfg.train.batch_size, batch_size_per_device=cfg.train.batch_size_per_device, world_size=1)
    assert n_accum == 2 and len(loader) == 4
    step = TrainStep(model, lr=cfg.train.lr, clip_grad=5.0)
    logs, n_samples = [], 0
    for epoch in range(3):
        n_samples = pretrain_one_epoch(step, loader, n_accum, 1, cfg, epoch, n_samples, log=logs.append)
    assert n_samples == 3 * 4 * 2 and step.optimizer.step_count == 3 * 4 // n_accum and len(logs) == 6
    # update iterations are i = 1, 3 of every epoch; the logged lr is the schedule at the fractional epoch i / 4 + epoch
    want = []
    for epoch in range(3):
        for i in (1, 3):
            s = i / 4 + epoch
            want.append(2e-3 * s / 1 if s < 1 else 1e-5 + (2e-3 - 1e-5) * 0.5 * (1 + math.cos(math.pi * (s - 1) / (3 - 1))))
    assert [round(lg["lr"], 10) for lg in logs] == [round(w, 10) for w in want]
    assert all(g["lr"] == pytest.approx(logs[-1]["lr"]) for g in step.optimizer.param_groups)
    assert [lg["n_samples"] for lg in logs] == [4, 8, 12, 16, 20, 24]
    assert {"loss", "grad_norm", "lr", "n_samples"} <= set(logs[0]) and any(k.endswith("mse_loss") for k in logs[0])
    assert all(torch.isfinite(lg["grad_norm"]) for lg in logs)
    assert float(logs[-1]["loss"]) < float(logs[0]["loss"])


def test_persistent_tables_are_not_allocated_from_a_recording_pool() -> None:
    """``hip.persistent`` / ``tape.const`` inside a recording: the table must not land on a block of the recording's private pool (the allocator would
    reuse the address of a freed temporary, and replaying the launches that wrote the temporary would overwrite the table)."""
    from cinema_amd import tape as T

    pool = torch.cuda.MemPool()
    K.RECORD = []
    try:
        with torch.cuda.use_mem_pool(pool):
            tmp = torch.empty(1 << 20, dtype=torch.float32, device=DEV)
            lo, hi = tmp.data_ptr(), tmp.data_ptr() + tmp.numel() * 4
            del tmp  # back to the pool: the next allocation of this size on this thread gets the same block
            again = torch.empty(1 << 20, dtype=torch.float32, device=DEV)
            assert again.data_ptr() == lo  # the hazard the helper avoids
            del again
            table = T.const(("test_persistent_table", 1 << 20), lambda: torch.arange(1 << 20, dtype=torch.float32).to(DEV))
            assert not (lo <= table.data_ptr() < hi)
            state = K.persistent(lambda: torch.zeros(1 << 20, dtype=torch.float32, device=DEV))
            assert not (lo <= state.data_ptr() < hi)
            assert float(table[12345]) == 12345.0
    finally:
        K.RECORD = None


# ---------------------------------------------------------------------------------------------------- error word of the in-launch split reductions
def _mfma_model():  # noqa: ANN202
    views = ["sax", "lax_2c"]
    kw = dict(image_size_dict={"sax": (64, 64, 8), "lax_2c": (64, 64)}, in_chans_dict=dict.fromkeys(views, 1), enc_patch_size_dict={"sax": (4, 4, 1), "lax_2c": (4, 4)},
              enc_scale_factor_dict={"sax": (2, 2, 1), "lax_2c": (2, 2)}, enc_conv_chans=[64, 128], enc_conv_n_blocks=1, enc_embed_dim=256, enc_depth=2, enc_n_heads=4,
              dec_embed_dim=128, dec_depth=2, dec_n_heads=4)
    torch.manual_seed(0)
    model = CineMA(**kw).to(DEV)
    gen = torch.Generator().manual_seed(1)
    batch = {v: torch.rand(2, 1, *kw["image_size_dict"][v], generator=gen).to(DEV) for v in views}
    return model, batch


def test_train_step_reads_the_split_reduction_error_word(tmp_path: Path) -> None:
    """A workgroup of the persistent GEMM (or of a split tail) that gives up its bounded wait for a partial tile sets an error word in the workspace
    (csrc/gemm256.hip, csrc/gemm.hip): the results of that launch are wrong.  ``TrainStep`` reads the words every ``check_every`` updates and
    ``save_checkpoint`` before writing: a planted word raises ``HipLibraryError`` at the next checked step (not before), the counter region is
    re-zeroed, training continues."""
    from cinema_amd.optim import GradScaler, TrainStep, save_checkpoint

    model, batch = _mfma_model()
    step = TrainStep(model, lr=1e-3, check_every=2)
    for _ in range(2):
        step(batch, 0.75)  # update 2 is a checked one: clean
    dev = torch.device("cuda", torch.cuda.current_device())
    ws = K._p256_workspace(dev)  # noqa: SLF001  (this stream's workspace: created here if the small model's groups ran elsewhere)
    words = ws.view(torch.int32)
    words[K.P256_ERROR_WORD] = 1
    step(batch, 0.75)  # update 3: not a checked one
    with pytest.raises(K.HipLibraryError, match="error word"):
        step(batch, 0.75)  # update 4: checked
    assert int(words[: K.P256_ERROR_WORD + 1].abs().sum()) == 0  # counters and the word are zero again
    for _ in range(2):
        loss, gnorm, _ = step(batch, 0.75)  # updates 5, 6 (checked): clean again
    assert math.isfinite(float(loss)) and math.isfinite(float(gnorm))
    tail = K._tail_counters(dev)  # noqa: SLF001
    tail[K.TAIL_ERROR_WORD] = 1
    with pytest.raises(K.HipLibraryError, match="split-tail"):
        save_checkpoint(tmp_path, 0, model, step.optimizer, GradScaler(), 0)
    assert int(tail.abs().sum()) == 0
    assert save_checkpoint(tmp_path, 0, model, step.optimizer, GradScaler(), 0).exists()


def test_hip_graph_step_smoke() -> None:
    """``TrainStep(hip_graph=True)``: forward + backward captured once and replayed as a HIP graph (kept for A/B against the recorded launch list).  The
    long-axis stream stays outside the capture: ``Tape.backward`` must not wait, from inside the capture, on an event of that non-capturing stream
    (ADVICE round 4).  Three optimisation steps, finite and falling loss; combining the graph with the fp8 delayed-scaling path is refused."""
    from cinema_amd import tape as T
    from cinema_amd.optim import TrainStep

    model, batch = _mfma_model()
    step = TrainStep(model, lr=1e-3, hip_graph=True)
    losses = [float(step(batch, 0.75)[0]) for _ in range(4)]
    torch.cuda.synchronize()
    assert all(math.isfinite(v) for v in losses) and losses[-1] < losses[0], losses
    prev = T.FP8_FORWARD
    T.FP8_FORWARD = True
    try:
        with pytest.raises(ValueError, match="hip_graph"):
            TrainStep(_mfma_model()[0], lr=1e-3, hip_graph=True)(batch, 0.75)
    finally:
        T.FP8_FORWARD = prev


# ---------------------------------------------------------------------------------------------------- module-level helpers of cinema.mae.mae (a20, a21)
def test_module_level_mse_loss_matches_the_reference_golden() -> None:
    """``from cinema.mae.mae import mse_loss`` on device tensors == the values the REFERENCE's ``mse_loss`` produced for the same target / prediction / mask
    (``tests/golden/layers.safetensors: mse/*``, written by oracle/make_golden.py from cinema/mae/mae.py:107-152), with and without target normalisation; the
    gradient with respect to the prediction is 2 (pred - target) / numel on the masked patches (rounded once to bf16, as the model's backward pass receives it)."""
    from cinema.mae.mae import mse_loss

    g = load_golden("layers.safetensors")
    target, pred, mask = g["mse/target"].to(DEV), g["mse/pred"].to(DEV), g["mse/mask"].to(DEV).bool()
    for nt in (0, 1):
        p = pred.clone().requires_grad_(True)
        loss, metrics = mse_loss(target, p, mask, bool(nt))
        assert abs(float(loss) - float(g[f"mse/{nt}/loss"])) <= 1e-5 * abs(float(g[f"mse/{nt}/loss"])) + 1e-7
        want = {k.split("/")[-1] for k in g if k.startswith(f"mse/{nt}/")} - {"loss"}
        assert set(metrics) == want, (set(metrics), want)
        for k in want:
            assert abs(float(metrics[k]) - float(g[f"mse/{nt}/{k}"])) <= 1e-5 * abs(float(g[f"mse/{nt}/{k}"])) + 1e-6, (nt, k)
        loss.backward()
        t = target
        if nt:
            t = (t - t.mean(-1, keepdim=True)) / (t.var(-1, keepdim=True) ** 0.5 + 1e-6)
        ref = 2 * (pred - t[mask].reshape(pred.shape)) / pred.numel()
        assert float(((p.grad - ref).abs() - 4e-3 * ref.abs()).max()) <= 1e-7  # the kernel hands the gradient over in bf16 (the operand of the head's backward GEMMs): one rounding


def test_module_level_add_pos_embed_and_append_mask_token() -> None:
    """``add_pos_embed_and_append_mask_token`` against the reference's own lines (cinema/mae/mae.py:91-104) evaluated with torch on the same tensors: concatenated
    and split forms, gradients of the visible tokens and of the mask token."""
    from cinema.mae.mae import add_pos_embed_and_append_mask_token

    torch.manual_seed(3)
    batch, n, n_keep, d = 3, 20, 7, 32
    mask = torch.stack([torch.randperm(n) >= n_keep for _ in range(batch)]).to(DEV)
    x = torch.randn(batch, n_keep, d, device=DEV)
    pe = torch.nn.Parameter(torch.randn(1, n, d, device=DEV), requires_grad=False)
    tok = torch.nn.Parameter(torch.randn(1, 1, d, device=DEV))

    def reference(xv: torch.Tensor, token: torch.Tensor) -> tuple:
        dec_pe = pe.expand(batch, -1, -1).contiguous()
        vis_pe = dec_pe[~mask].reshape(batch, n_keep, d)
        mask_pe = dec_pe[mask].reshape(batch, n - n_keep, d)
        return xv + vis_pe, token + mask_pe

    xr, tr = x.clone().requires_grad_(True), tok.detach().clone().requires_grad_(True)
    rv, rm = reference(xr, tr)
    xa = x.clone().requires_grad_(True)
    out = add_pos_embed_and_append_mask_token(xa, mask, pe, tok, concat=True)
    assert out.shape == (batch, n, d)
    assert torch.allclose(out, torch.cat([rv, rm], dim=1), atol=1e-6)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    (torch.cat([rv, rm], dim=1) * w).sum().backward()
    assert torch.allclose(xa.grad, xr.grad, atol=1e-6) and torch.allclose(tok.grad, tr.grad, atol=1e-4)
    v2, m2 = add_pos_embed_and_append_mask_token(x, mask, pe, tok, concat=False)
    assert torch.allclose(v2, rv, atol=1e-6) and torch.allclose(m2, rm, atol=1e-6)
