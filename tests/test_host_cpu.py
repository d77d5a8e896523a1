"""CPU-side checks (no GPU needed): the C-ABI library loads and exports every symbol the header declares, the host-side
mirror of the reference interface (state_dict layout, seeded construction, schedules, masks, error behaviour) and the
multi-process gradient synchroniser over gloo."""

from __future__ import annotations

import ctypes
import json
import math
import os
import re
import sys
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT, load_golden

import cinema_oracle as O  # noqa: E402
from cinema_amd import CineMA, hip, patchify, unpatchify  # noqa: E402
from cinema_amd.config import to_config  # noqa: E402
from cinema_amd.convvit import TokenSelection, upsample_mask  # noqa: E402
from cinema_amd.mae.mae import get_batch_random_patch_mask, get_decoder_patch_size, get_model  # noqa: E402
from cinema_amd.optim import FlatModel, adjust_learning_rate, get_n_accum_steps, param_groups_weight_decay  # noqa: E402
from cinema_amd.vit import get_pos_embed, get_vit_config  # noqa: E402


# ---------------------------------------------------------------------------------------------------- C-ABI
def test_library_exports_every_declared_symbol() -> None:
    header = (ROOT / "include" / "cinema_hip.h").read_text()
    declared = set(re.findall(r"^(?:int|long long) (cinema_\w+)\(", header, flags=re.M))
    assert declared, "no declarations parsed from include/cinema_hip.h"
    assert declared == set(hip.EXPORTED_SYMBOLS), declared ^ set(hip.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(str(hip.library_path()))
    for name in declared:
        assert hasattr(lib, name), f"libcinema_hip.so does not export {name}"
    out = (ctypes.c_int * 8)()
    assert lib.cinema_hip_info(out) == 0 and out[0] == 1  # abi version; no compute call


def test_gemm_args_struct_matches_header_field_order() -> None:
    header = (ROOT / "include" / "cinema_hip.h").read_text()
    body = header[header.index("typedef struct {"):header.index("} cinema_gemm_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.replace("typedef struct {", "").strip()
        if not decl:
            continue
        names = decl.split(",")
        first = names[0].split()[-1].lstrip("*")
        fields.append(first)
        fields += [n.strip().lstrip("*") for n in names[1:]]
    assert fields == [f[0] for f in hip.GemmArgs._fields_]  # noqa: SLF001


def test_product_path_never_imports_the_oracle() -> None:
    for path in (ROOT / "cinema_amd").rglob("*.py"):
        text = path.read_text()
        assert "cinema_oracle" not in text and "oracle" not in re.findall(r"^\s*(?:from|import)\s+(\w+)", text, flags=re.M), path


def test_cpu_tensors_fail_loudly() -> None:
    with pytest.raises(hip.HipLibraryError):
        hip.gemm(torch.zeros(8, 8, dtype=torch.bfloat16), torch.zeros(8, 8, dtype=torch.bfloat16))
    model = CineMA(**mini_kwargs())
    with pytest.raises(hip.HipLibraryError):
        model({"sax": torch.rand(1, 1, 32, 32, 4)}, 0.75)


# ---------------------------------------------------------------------------------------------------- module layer
def mini_kwargs() -> dict:
    views = ["sax", "lax_2c", "lax_3c", "lax_4c"]
    return dict(image_size_dict={v: (32, 32, 4) if v == "sax" else (32, 32) for v in views}, in_chans_dict=dict.fromkeys(views, 1),
                enc_patch_size_dict={v: (4, 4, 1) if v == "sax" else (4, 4) for v in views},
                enc_scale_factor_dict={v: (2, 2, 1) if v == "sax" else (2, 2) for v in views}, enc_conv_chans=[16, 32], enc_conv_n_blocks=1,
                enc_embed_dim=64, enc_depth=2, enc_n_heads=4, dec_embed_dim=32, dec_depth=2, dec_n_heads=4)


def base_kwargs(size: str, sax: tuple, lax: tuple) -> dict:
    views = ["sax", "lax_2c", "lax_3c", "lax_4c"]
    return dict(image_size_dict={v: sax if v == "sax" else lax for v in views}, in_chans_dict=dict.fromkeys(views, 1),
                enc_patch_size_dict={v: (4, 4, 1) if v == "sax" else (4, 4) for v in views},
                enc_scale_factor_dict={v: (2, 2, 1) if v == "sax" else (2, 2) for v in views}, enc_conv_chans=[64, 128], enc_conv_n_blocks=2,
                **get_vit_config(size))


@pytest.mark.parametrize(("name", "size", "sax", "lax"), [("base_4view_192", "base", (192, 192, 16), (192, 192)),
                                                         ("large_4view_256", "large", (256, 256, 24), (256, 256))])
def test_state_dict_layout_matches_reference_manifest(name: str, size: str, sax: tuple, lax: tuple) -> None:
    man = json.loads((GOLDEN / "state_dict_manifests.json").read_text())[name]
    with torch.device("meta"):
        model = CineMA(**base_kwargs(size, sax, lax))
    sd = model.state_dict()
    assert list(sd) == list(man["keys"])
    assert {k: list(v.shape) for k, v in sd.items()} == man["keys"]
    assert sum(p.numel() for p in model.parameters()) == man["n_params"]
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == man["n_trainable"]
    groups = param_groups_weight_decay(model, 0.05)
    names = {id(p): k for k, p in model.named_parameters()}
    assert sorted(names[id(p)] for p in groups[0]["params"]) == sorted(man["no_decay"])
    assert "encoder.cls_token" in {names[id(p)] for p in groups[1]["params"]}  # tokens ARE decayed (pretrain.py:365)


@pytest.mark.parametrize("name", ["tiny_sax", "mini_4view"])
def test_seeded_construction_is_identical_to_the_reference(name: str) -> None:
    fp = json.loads((GOLDEN / f"{name}_init_fingerprint.json").read_text())
    kw = mini_kwargs() if name == "mini_4view" else dict(
        image_size_dict={"sax": (128, 128, 8)}, in_chans_dict={"sax": 1}, enc_patch_size_dict={"sax": (4, 4, 1)},
        enc_scale_factor_dict={"sax": (2, 2, 1)}, enc_conv_chans=[64, 128], enc_conv_n_blocks=2, **get_vit_config("tiny"))
    torch.manual_seed(fp["seed"])
    sd = CineMA(**kw).state_dict()
    assert list(sd) == list(fp["params"])
    for k, ref in fp["params"].items():
        assert list(sd[k].shape) == ref["shape"], k
        assert float(sd[k].double().sum()) == pytest.approx(ref["sum"], rel=1e-6, abs=1e-7), k
        assert [float(x) for x in sd[k].flatten()[:4]] == pytest.approx(ref["head"], rel=1e-6, abs=1e-8), k


def test_state_dict_round_trip_with_reference_weights() -> None:
    g = load_golden("mini_4view.safetensors")
    params = {k[len("param/"):]: v for k, v in g.items() if k.startswith("param/")}
    model = CineMA(**mini_kwargs())
    missing, unexpected = model.load_state_dict(params, strict=True)
    assert not missing and not unexpected
    for k, v in model.state_dict().items():
        assert torch.equal(v, params[k]), k


def test_get_model_config_mapping_and_attributes() -> None:
    cfg = to_config({"grad_ckpt": True, "data": {"sax": {"patch_size": [64, 64, 8], "in_chans": 1}, "lax": {"patch_size": [64, 64], "in_chans": 1}},
                     "model": {"size": "tiny", "patch_size": [4, 4, 1], "scale_factor": [2, 2, 1], "enc_conv_chans": [16, 32], "enc_conv_n_blocks": 1}})
    model = get_model(cfg)
    assert model.views == ["sax", "lax_2c", "lax_3c", "lax_4c"] and model.grad_ckpt is True
    assert model.enc_down_dict["sax"].patch_embed.grid_size == (4, 4, 8) and model.enc_down_dict["lax_2c"].patch_embed.n_patches == 16
    assert model.dec_patch_size_dict == {"sax": (16, 16, 1), "lax_2c": (16, 16), "lax_3c": (16, 16), "lax_4c": (16, 16)}
    assert len(model.encoder.blocks) == 1 and model.cross_attn and not model.norm_target
    assert get_decoder_patch_size((192, 192, 16), 2, (4, 4, 1), (2, 2, 1)) == (16, 16, 1)
    with pytest.raises(ValueError):
        get_vit_config("giant")


def test_patchify_tables_and_masks_against_reference_vectors() -> None:
    g = load_golden("layers.safetensors")
    assert torch.equal(patchify(g["patchify/image3d"], (2, 3, 1)), g["patchify/out3d"])
    assert torch.equal(patchify(g["patchify/image2d"], (2, 2)), g["patchify/out2d"])
    assert torch.equal(unpatchify(g["patchify/out3d"], (2, 3, 1), (2, 2, 2)), g["patchify/image3d"])
    assert torch.equal(unpatchify(g["patchify/out2d"], (2, 2), (2, 3)), g["patchify/image2d"])
    with pytest.raises(ValueError):
        patchify(torch.zeros(1, 1, 5, 4), (2, 2))
    with pytest.raises(ValueError):
        unpatchify(torch.zeros(1, 4, 8), (2, 2), (2, 3))
    for name, dim, grid in [("sax768", 768, (12, 12, 16)), ("lax768", 768, (12, 12)), ("odd", 20, (2, 3, 4)), ("odd2d", 10, (3, 2))]:
        pe = get_pos_embed(dim, grid)
        assert not pe.requires_grad
        pe = pe[:, ::37] if pe.shape[1] > 64 else pe
        assert torch.allclose(pe, g[f"pos_embed/{name}"], atol=1e-6), name
    for i in range(4):
        out = upsample_mask(g[f"upsample_mask/{i}/in"].bool(), tuple(g[f"upsample_mask/{i}/scale"].tolist()))
        assert torch.equal(out, g[f"upsample_mask/{i}/out"].bool())
    for n, r in [(16, 0.75), (10, 0.5), (7, 0.3), (512, 0.75), (2304, 0.75)]:
        mask = get_batch_random_patch_mask(3, n, r, torch.device("cpu"))
        assert mask.shape == (3, n) and int((~mask).sum()) == int(n * (1 - r)) * 3  # exact keep count (mae_test.py:30-32)
    assert not get_batch_random_patch_mask(2, 5, 0.0, torch.device("cpu")).any()
    with pytest.raises(ValueError):
        get_batch_random_patch_mask(2, 5, -0.1, torch.device("cpu"))


def test_token_selection_is_raster_order_like_boolean_indexing() -> None:
    mask = get_batch_random_patch_mask(4, 48, 0.75, torch.device("cpu"))
    sel = TokenSelection(mask, 4, 48, torch.device("cpu"), n_masked=36)
    flat = torch.arange(4 * 48).reshape(4, 48)
    assert torch.equal(sel.keep.long(), flat[~mask]) and torch.equal(sel.drop.long(), flat[mask])
    assert torch.equal(sel.keep_pos.long(), flat[~mask] % 48) and (sel.n_keep, sel.n_drop) == (12, 36)
    sel2 = TokenSelection(mask, 4, 48, torch.device("cpu"))  # count read back from the mask
    assert sel2.n_drop == 36
    full = TokenSelection(None, 2, 5, torch.device("cpu"))
    assert full.all_tokens and full.keep.tolist() == list(range(10))


def test_lr_schedule_accumulation_and_reference_errors() -> None:
    for row in json.loads((GOLDEN / "lr_schedule.json").read_text()):
        opt = torch.optim.SGD([{"params": [torch.zeros(1, requires_grad=True)]}, {"params": [torch.zeros(1, requires_grad=True)], "lr_scale": 0.5}], lr=0.1)
        assert adjust_learning_rate(opt, *row["args"]) == pytest.approx(row["lr"], rel=1e-12, abs=1e-18)
        assert [g["lr"] for g in opt.param_groups] == pytest.approx(row["group_lrs"], rel=1e-12, abs=1e-18)
    assert get_n_accum_steps(64, 16, 2) == 2 and get_n_accum_steps(64, 16, 4) == 1
    with pytest.raises(ValueError):
        get_n_accum_steps(16, 16, 2)
    with pytest.raises(ValueError):
        get_n_accum_steps(48, 16, 2)
    model = CineMA(**mini_kwargs())
    with pytest.raises(ValueError):
        model({"bogus": torch.zeros(1, 1, 32, 32)}, 0.75)


def test_flat_model_views_share_storage() -> None:
    model = CineMA(**mini_kwargs())
    before = {k: v.clone() for k, v in model.state_dict().items()}
    flat = FlatModel(model, 0.05)
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k]), k
    n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    assert flat.numel >= n_train and flat.numel - n_train < 4 * len(list(model.parameters()))
    p = model.encoder.blocks[0].attn.q.weight
    flat.flat_grad.fill_(2.0)
    assert float(p.grad.sum()) == 2.0 * p.numel()
    flat.flat_param.zero_()
    assert float(p.abs().sum()) == 0.0
    (a0, b0), (a1, b1) = flat.ranges
    assert a0 == 0 and b0 == a1 and b1 == flat.numel  # [no-decay | decay] contiguous ranges


# ---------------------------------------------------------------------------------------------------- N > 1 path on gloo
def _ddp_worker(rank: int, world: int, port: int, tmp: str) -> None:
    sys.path.insert(0, str(ROOT))
    from cinema_amd.ddp import GradientSynchronizer, ddp_setup

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    ddp_setup(rank, world, port=port, backend="gloo")
    torch.manual_seed(100 + rank)  # different initial weights per rank: the broadcast must make them equal to rank 0's
    model = CineMA(**mini_kwargs())
    flat = FlatModel(model, 0.05)
    sync = GradientSynchronizer(world, bucket_bytes=64 << 10)  # small buckets -> several collectives
    sync.attach(flat)
    assert len(sync.buckets) > 1
    torch.save(flat.flat_param.clone(), f"{tmp}/param{rank}.pt")
    flat.flat_grad.copy_(torch.arange(flat.numel, dtype=torch.float32) * (rank + 1))
    sync.all_reduce()
    torch.save(flat.flat_grad.clone(), f"{tmp}/grad{rank}.pt")
    # overlapped path: two encoder blocks report "gradients complete" during the backward pass (a non-direct parameter set is refused),
    # the remainder goes out at the end; every element must be reduced exactly once
    from types import SimpleNamespace

    flat.flat_grad.copy_(torch.arange(flat.numel, dtype=torch.float32) * (rank + 1))
    blocks = [list(model.encoder.blocks[i].parameters()) for i in (1, 0)]
    fake_tape = SimpleNamespace(pvars={id(p): SimpleNamespace(direct=True) for b in blocks for p in b})
    sync.min_early = 64  # the mini model's blocks are tiny
    sync.arm(True)
    for b in blocks:
        sync.params_done(fake_tape, b)
    assert len(sync._early) >= 2  # noqa: SLF001
    n_early = len(sync._early)  # noqa: SLF001
    head = list(model.decoder.blocks[0].parameters())
    sync.params_done(SimpleNamespace(pvars={id(p): SimpleNamespace(direct=False) for p in head}), head)
    assert len(sync._early) == n_early  # noqa: SLF001
    sync.all_reduce()
    torch.save(flat.flat_grad.clone(), f"{tmp}/grad_overlap{rank}.pt")
    sync.arm(False)  # accumulation micro-step: the hook must stay silent
    sync.params_done(fake_tape, blocks[0])
    assert not sync._early and not sync._works  # noqa: SLF001
    # the same overlapped schedule as explicit reduce-scatter + all-gather on the flat ranges (algorithm="rs_ag"): two collectives per range, same payload,
    # same mean; a range whose length is not a multiple of the world size takes a small all-reduce for the remainder
    rs = GradientSynchronizer(world, bucket_bytes=64 << 10, algorithm="rs_ag")
    rs.attach(flat)
    rs.min_early = 64
    flat.flat_grad.copy_(torch.arange(flat.numel, dtype=torch.float32) * (rank + 1))
    rs.arm(True)
    for b in blocks:
        rs.params_done(fake_tape, b)
    rs.all_reduce()
    torch.save({"grad": flat.flat_grad.clone(), "n": rs.n_collectives_last, "n_ar": sync.n_collectives_last, "bytes": (rs.bytes_last, sync.bytes_last)}, f"{tmp}/grad_rsag{rank}.pt")
    odd = torch.arange(11, dtype=torch.float32) * (rank + 1)  # 11 = 2 * 5 + 1: shard of 5 per rank + a remainder of 1
    rs.arm(True)
    rs._launch(odd)  # noqa: SLF001
    for w in rs._works:  # noqa: SLF001
        w.wait()
    torch.save(odd / world, f"{tmp}/odd_rsag{rank}.pt")
    ok = sync.all_finite(torch.tensor(float("nan") if rank == 1 else 1.0))
    torch.save(ok, f"{tmp}/finite{rank}.pt")
    torch.distributed.destroy_process_group()


def test_gradient_synchronizer_world_size_2_gloo(tmp_path: Path) -> None:
    from cinema_amd.ddp import get_free_port

    mp.spawn(_ddp_worker, args=(2, get_free_port(), str(tmp_path)), nprocs=2, join=True)
    p0, p1 = torch.load(tmp_path / "param0.pt"), torch.load(tmp_path / "param1.pt")
    assert torch.equal(p0, p1)  # rank 0's parameters everywhere
    g0, g1 = torch.load(tmp_path / "grad0.pt"), torch.load(tmp_path / "grad1.pt")
    expect = torch.arange(g0.numel(), dtype=torch.float32) * 1.5  # mean of 1x and 2x
    assert torch.equal(g0, g1) and torch.allclose(g0, expect)
    for r in (0, 1):
        assert torch.equal(torch.load(tmp_path / f"grad_overlap{r}.pt"), g0)  # overlapped schedule == plain bucketed schedule
        rs = torch.load(tmp_path / f"grad_rsag{r}.pt")
        assert torch.equal(rs["grad"], g0)  # reduce-scatter + all-gather == all-reduce (two ranks: one addition per element, no order freedom)
        assert rs["n"] == 2 * rs["n_ar"] and rs["bytes"][0] == rs["bytes"][1] > 0
        assert torch.equal(torch.load(tmp_path / f"odd_rsag{r}.pt"), torch.arange(11, dtype=torch.float32) * 1.5)
    assert float(torch.load(tmp_path / "finite0.pt")) == 0.0 and float(torch.load(tmp_path / "finite1.pt")) == 0.0  # collective NaN decision


def _ddp_bf16_worker(rank: int, world: int, port: int, tmp: str) -> None:
    sys.path.insert(0, str(ROOT))
    from cinema_amd.ddp import GradientSynchronizer, ddp_setup

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    ddp_setup(rank, world, port=port, backend="gloo")
    torch.manual_seed(0)
    model = CineMA(**mini_kwargs())
    flat = FlatModel(model, 0.05)
    gen = torch.Generator().manual_seed(50 + rank)
    grad = torch.randn(flat.numel, generator=gen) * torch.logspace(-4, 0, flat.numel)  # four decades of magnitudes, different per rank
    out = {}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        sync = GradientSynchronizer(world, bucket_bytes=64 << 10, exchange_dtype=dt, overlap=False)
        sync.attach(flat)
        flat.flat_grad.copy_(grad)
        sync.arm(True)
        sync.all_reduce()
        out[name] = (flat.flat_grad.clone(), sync.bytes_last)
    # gradient accumulation: n_accum_steps = 2 -> the first micro-step must not communicate (reference pretrain.py:259-269 all-reduces on every
    # micro-step through DDP; here only the boundary step does), the second one must
    sync = GradientSynchronizer(world, bucket_bytes=64 << 10, overlap=True, min_early_bytes=256)
    sync.attach(flat)
    from types import SimpleNamespace

    blocks = [list(model.encoder.blocks[i].parameters()) for i in (1, 0)]
    fake_tape = SimpleNamespace(pvars={id(p): SimpleNamespace(direct=True) for b in blocks for p in b})
    counts = []
    for micro in range(4):
        update = (micro + 1) % 2 == 0
        flat.flat_grad.add_(grad) if micro % 2 else flat.flat_grad.copy_(grad)
        sync.arm(update)
        for b in blocks:  # the backward pass reports the blocks on every micro-step
            sync.params_done(fake_tape, b)
        if update:
            sync.all_reduce()
        counts.append(sync.n_collectives_total)
    torch.save({"f32": out["f32"][0], "bf16": out["bf16"][0], "bytes": (out["f32"][1], out["bf16"][1]), "counts": counts, "accum": flat.flat_grad.clone()},
               f"{tmp}/bf16_{rank}.pt")
    torch.distributed.destroy_process_group()


def test_bf16_gradient_exchange_and_accumulation_window_world_size_2_gloo(tmp_path: Path) -> None:
    """(a) The optional bf16 exchange: half the payload bytes, every rank ends with bit-identical gradients, relative L2 error against the fp32
    exchange <= 1e-2 (two roundings to bf16: 2^-9 each).  (b) With n_accum_steps = 2 only the boundary micro-step communicates: the collective counter
    stands still on the accumulation micro-step although the backward hooks fire, and the exchanged buffer holds the mean of the accumulated gradients."""
    from cinema_amd.ddp import get_free_port

    mp.spawn(_ddp_bf16_worker, args=(2, get_free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "bf16_0.pt"), torch.load(tmp_path / "bf16_1.pt")
    assert torch.equal(r0["f32"], r1["f32"]) and torch.equal(r0["bf16"], r1["bf16"])  # replicas bit-identical in both modes
    rel = float((r0["bf16"] - r0["f32"]).norm() / r0["f32"].norm())
    assert 0.0 < rel <= 1e-2, rel
    assert r0["bytes"][1] * 2 == r0["bytes"][0] > 0
    c = r0["counts"]
    assert c[0] == 0 and c[1] > 0 and c[2] == c[1] and c[3] == 2 * c[1], c  # micro-steps 0 and 2 accumulate silently
    assert r0["counts"] == r1["counts"] and torch.equal(r0["accum"], r1["accum"])


def test_oracle_two_rank_average_equals_full_batch_gradient() -> None:
    """Data-parallel semantics the synchroniser implements: mean over ranks of per-rank mean-loss gradients == full-batch gradient
    (equal per-rank batch sizes, same masks), checked with the CPU oracle on the mini config."""
    g = load_golden("mini_4view.safetensors")
    params = {k[len("param/"):]: v for k, v in g.items() if k.startswith("param/")}
    images = {k[len("image/"):]: v for k, v in g.items() if k.startswith("image/")}
    masks = {k[len("mask/"):]: v.bool() for k, v in g.items() if k.startswith("mask/")}
    cfg = O.MAEConfig(**mini_kwargs())

    def grads(sl: slice) -> dict:
        p = {k: v.clone().requires_grad_(not k.endswith("pos_embed")) for k, v in params.items()}
        loss, _, _ = O.mae_forward(p, cfg, {k: v[sl] for k, v in images.items()}, {k: v[sl] for k, v in masks.items()})
        loss.backward()
        return {k: v.grad for k, v in p.items() if v.grad is not None}

    full, r0, r1 = grads(slice(0, 2)), grads(slice(0, 1)), grads(slice(1, 2))
    for k in full:
        assert torch.allclose(full[k], 0.5 * (r0[k] + r1[k]), rtol=1e-4, atol=1e-7), k
    assert math.isfinite(float(sum(v.abs().sum() for v in full.values())))


# ---------------------------------------------------------------------------------------------------- ConvViT host logic
def _convvit_kwargs() -> dict:
    kw = json.loads((GOLDEN / "convvit_meta.json").read_text())["kwargs"]
    for key in ("image_size_dict", "enc_patch_size_dict", "enc_scale_factor_dict"):
        kw[key] = {v: tuple(s) for v, s in kw[key].items()}
    return kw


def test_convvit_seeded_construction_and_state_dict_match_the_reference() -> None:
    from cinema_amd.convvit import ConvViT

    g = load_golden("convvit_mini.safetensors")
    ref = {k[len("param/"):]: v for k, v in g.items() if k.startswith("param/")}
    torch.manual_seed(0)
    sd = ConvViT(**_convvit_kwargs()).state_dict()
    assert set(sd) == set(ref)  # heads are created after apply(init_weights) (convvit.py:439-445)
    for k, v in sd.items():
        assert torch.equal(v, ref[k]), k  # same RNG draw order -> bit-identical initial weights, heads with torch's default init


def test_convvit_param_groups_lr_decay_match_the_reference() -> None:
    from cinema_amd.convvit import ConvViT, get_layer_id_for_vit, param_groups_lr_decay

    meta = json.loads((GOLDEN / "convvit_meta.json").read_text())
    model = ConvViT(**_convvit_kwargs())
    groups = param_groups_lr_decay(model, no_weight_decay_list=[], weight_decay=0.05, layer_decay=0.75)
    names = {id(p): n for n, p in model.named_parameters()}
    got = [{"lr_scale": g["lr_scale"], "weight_decay": g["weight_decay"], "params": [names[id(p)] for p in g["params"]]} for g in groups]
    assert got == meta["param_groups"]
    assert get_layer_id_for_vit("encoder.blocks.1.attn.q.weight", 3) == 2 and get_layer_id_for_vit("enc_down_dict.sax.linear.weight", 3) == 0
    assert get_layer_id_for_vit("pred_head_dict.cls.weight", 3) == 3 and get_layer_id_for_vit("encoder.cls_token", 3) == 0


def test_convvit_load_pretrain_weights_matches_the_reference(tmp_path: Path) -> None:
    from safetensors.torch import save_file

    from cinema_amd.convvit import ConvViT, load_pretrain_weights

    meta = json.loads((GOLDEN / "convvit_meta.json").read_text())["load_pretrain"]
    ck = GOLDEN / "convvit_mae_ckpt.safetensors"
    mae_sd = load_golden("convvit_mae_ckpt.safetensors")
    torch.manual_seed(2)
    fresh = ConvViT(**_convvit_kwargs())
    before = {k: v.detach().clone() for k, v in fresh.state_dict().items()}
    loaded = load_pretrain_weights(fresh, views=["sax", "lax_2c"], ckpt_path=ck, freeze=True)
    after = loaded.state_dict()
    assert sorted(k for k in after if not torch.equal(after[k], before[k])) == meta["changed"]
    assert sorted(n for n, p in loaded.named_parameters() if not p.requires_grad) == meta["frozen"]
    w = after["enc_down_dict.sax.conv_blocks.0.patch_embed.conv.weight"]  # single-frame filter tiled over the 2 frames
    assert torch.equal(w[:, 0], w[:, 1]) and torch.equal(w[:, :1], mae_sd["enc_down_dict.sax.conv_blocks.0.patch_embed.conv.weight"])
    with pytest.raises(ValueError, match="Missing keys from checkpoint"):  # a 2-view model cannot be filled from one view (reference behaviour)
        load_pretrain_weights(ConvViT(**_convvit_kwargs()), views=["sax"], ckpt_path=ck, freeze=False)
    bad = dict(mae_sd)
    bad["encoder.not_a_key"] = torch.zeros(1)
    save_file(bad, str(tmp_path / "bad.safetensors"))
    with pytest.raises(ValueError, match="Unexpected keys"):
        load_pretrain_weights(ConvViT(**_convvit_kwargs()), views=["sax", "lax_2c"], ckpt_path=tmp_path / "bad.safetensors", freeze=False)


def test_convvit_api_errors_on_cpu() -> None:
    from cinema_amd.convvit import ConvViT

    model = ConvViT(**_convvit_kwargs())
    with pytest.raises(ValueError):
        model.feature_forward({"bogus": torch.zeros(1, 2, 32, 32)}, None)
    with pytest.raises(NotImplementedError):
        model({"sax": torch.zeros(1, 2, 32, 32, 4), "lax_2c": torch.zeros(1, 2, 32, 32)}, None, reduce="none")
    with pytest.raises(hip.HipLibraryError):  # no CPU fallback
        model({"sax": torch.zeros(1, 2, 32, 32, 4), "lax_2c": torch.zeros(1, 2, 32, 32)}, None, reduce="cls")


# ---------------------------------------------------------------------------------------------------- ConvUNetR host logic
def _unetr_kwargs() -> dict:
    kw = json.loads((GOLDEN / "convunetr_meta.json").read_text())["kwargs"]
    for key in ("image_size_dict", "enc_patch_size_dict", "enc_scale_factor_dict", "dec_patch_size_dict", "dec_scale_factor_dict"):
        kw[key] = {v: tuple(s) for v, s in kw[key].items()}
    kw["dec_chans"] = tuple(kw["dec_chans"])
    return kw


def test_convunetr_state_dict_seeded_init_and_compat_checks_match_the_reference() -> None:
    from cinema_amd.segmentation.convunetr import ConvUNetR, check_conv_unetr_enc_dec_compatiblity

    meta = json.loads((GOLDEN / "convunetr_meta.json").read_text())
    g = load_golden("convunetr_mini.safetensors")
    ref = {k[len("param/"):]: v for k, v in g.items() if k.startswith("param/")}
    torch.manual_seed(0)
    model = ConvUNetR(**_unetr_kwargs())
    sd = model.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == meta["state_dict"]
    for k, v in sd.items():
        assert torch.equal(v, ref[k]), k  # same RNG draw order as the reference constructor
    assert model.n_layers_wo_skip == meta["n_layers_wo_skip"] and len(model.dec_down_blocks_dict["sax"]) == meta["n_downsample_layers"]
    assert list(check_conv_unetr_enc_dec_compatiblity((4, 4, 1), (2, 2, 1), 2, 5, (2, 2, 1), (2, 2, 1))) == meta["compat"]["acdc"]
    with pytest.raises(ValueError, match="must be less than dec_depth"):
        check_conv_unetr_enc_dec_compatiblity((4, 4, 1), (2, 2, 1), 5, 5, (2, 2, 1), (2, 2, 1))
    with pytest.raises(ValueError, match="must be greater than dec_patch_size"):
        check_conv_unetr_enc_dec_compatiblity((2, 2, 1), (2, 2, 1), 2, 5, (4, 4, 1), (2, 2, 1))
    with pytest.raises(ValueError, match="must be equal to"):
        check_conv_unetr_enc_dec_compatiblity((6, 6, 1), (2, 2, 1), 2, 5, (2, 2, 1), (2, 2, 1))
    with pytest.raises(ValueError):
        model({"bogus": torch.zeros(1, 1, 64, 64)})


def test_tape_host_and_const_entries_of_a_recording() -> None:
    """Host logic of recorded steps (cinema_amd/replay.py) without a GPU: tape.const caches by key; tape.host is a plain call outside a
    recording and, inside one, returns static tensors plus a host entry that recomputes them IN PLACE from the current inputs."""
    from cinema_amd import hip as K
    from cinema_amd import tape as T

    calls = []

    def table() -> torch.Tensor:
        calls.append(1)
        return torch.arange(5, dtype=torch.int32)

    a, b = T.const(("test_table", 5), table), T.const(("test_table", 5), table)
    assert a is b and len(calls) == 1
    mask = torch.tensor([[True, False, True, False]])

    def select() -> tuple:
        return ((~mask).nonzero()[:, 1].to(torch.int32), mask.nonzero()[:, 1].to(torch.int32))

    keep, drop = T.host(select)  # eager: no recording
    assert keep.tolist() == [1, 3] and drop.tolist() == [0, 2] and not T.recording()
    K.RECORD = rec = []
    try:
        keep, drop = T.host(select)
        assert T.recording() and len(rec) == 1 and rec[0][0] is None
        ptrs = (keep.data_ptr(), drop.data_ptr())
        mask.copy_(torch.tensor([[False, False, True, True]]))  # "next step": new mask in the same static tensor
        rec[0][1]()
        assert keep.tolist() == [0, 1] and drop.tolist() == [2, 3] and (keep.data_ptr(), drop.data_ptr()) == ptrs
    finally:
        K.RECORD = None


def test_segmentation_model_builder_maps_the_acdc_config() -> None:
    """``get_segmentation_model`` (reference ``cinema/segmentation/train.py:31-75``) with the ACDC recipe's model section (``acdc/config.yaml:56-66``)."""
    from cinema_amd.config import to_config
    from cinema_amd.segmentation.train import get_segmentation_model

    cfg = to_config({"grad_ckpt": True, "data": {"sax": {"patch_size": [64, 64, 4], "in_chans": 1, "spacing": [1.0, 1.0, 10.0]}},
                     "model": {"name": "convunetr", "views": "sax", "out_chans": 4,
                               "convunetr": {"size": "tiny", "enc_patch_size": [4, 4, 1], "enc_scale_factor": [2, 2, 1], "enc_conv_chans": [8, 16], "enc_conv_n_blocks": 1,
                                             "dec_chans": [4, 8, 16, 32, 64], "dec_patch_size": [2, 2, 1], "dec_scale_factor": [2, 2, 1], "dropout": 0.1,
                                             "drop_path": 0.1}}})
    model = get_segmentation_model(cfg)
    assert type(model).__name__ == "ConvUNetR" and model.grad_ckpt is True
    cfg.model.name = "unet"
    with pytest.raises(ValueError):
        get_segmentation_model(cfg)


def test_hot_kernels_have_no_scratch() -> None:
    """Register spills are silent and cost 2-5x in an MFMA loop (round 3: a reordered reduction made all 24 instances of the persistent GEMM spill 60-84
    registers without any test noticing): the code-object metadata of the built library must show zero scratch for every GEMM, attention, LayerNorm,
    sparse-stem and optimiser kernel.  Known, measured exception: the general epilogue class is not on the step's path."""
    import sys

    llvm = Path("/opt/rocm/lib/llvm/bin")
    if not (llvm / "llvm-readelf").exists() or not (llvm / "clang-offload-bundler").exists():
        pytest.skip("ROCm LLVM tools not installed")
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    from kernel_resources import kernel_table

    rows = kernel_table()
    assert len(rows) > 200
    hot = ("gemm_p256", "gemm_mfma", "gemm_fp8", "gemm_conv", "attn_", "ln_fwd", "ln_bwd", "sparse_dwconv", "adamw", "tail_fixup", "splitk_reduce")
    bad = [(r["name"][:80], r["private_segment_fixed_size"]) for r in rows if any(h in r["name"] for h in hot) and int(r["private_segment_fixed_size"]) > 0]
    assert not bad, bad


def test_fp8_sites_belong_to_the_parameter_object_and_slots_are_recycled() -> None:
    """Delayed-scaling sites hang on the parameter OBJECT (round 4 keyed them by id(parameter): a new model re-using a dead parameter's id inherited a
    ready site with a foreign scale, and a few models exhausted the 1024 slots): a site is ready only after an update that follows ITS creation, a dead
    parameter's slots are handed out again as fresh (not ready) sites, and the registry never grows past the live sites."""
    import gc

    from cinema_amd import tape as T

    reg = T.Fp8Sites(torch.device("cpu"))
    reg.update = lambda: setattr(reg, "updates", reg.updates + 1)  # (the real update is one HIP launch: host logic only here)
    a = torch.nn.Parameter(torch.zeros(4))
    sa, sb = reg.site(a, "x"), reg.site(a, "dy")
    assert reg.site(a, "x") is sa and sa is not sb and not sa.ready and not reg.all_ready()
    reg.update()
    assert sa.ready and sb.ready and reg.all_ready()
    slots_a = sorted(a._cinema_q8_slots[id(reg)])  # noqa: SLF001  (one slot list per registry)
    del a, sa, sb
    gc.collect()
    assert not reg.live and sorted(reg.free) == slots_a
    b = torch.nn.Parameter(torch.zeros(4))  # may or may not re-use the id: either way its sites start uncalibrated, in the recycled slots
    s1 = reg.site(b, "x")
    assert not s1.ready and reg.n_alloc == 2 and len(reg.live) == 1 and float(s1.scale) == 1.0
    reg.update()
    assert s1.ready
    keep = [torch.nn.Parameter(torch.zeros(1)) for _ in range(40)]
    for _ in range(30):  # 30 generations x 40 parameters x 2 sites = 2400 sites through 1024 slots
        gen = [torch.nn.Parameter(torch.zeros(1)) for _ in range(40)]
        for q in gen:
            reg.site(q, "x"), reg.site(q, "dy")
        del gen, q
        gc.collect()
    assert reg.n_alloc <= 2 + 80 and len(reg.live) == 1 and keep
    # the same parameter under TWO registries (a second device / a second Fp8Sites): each registry gets its own slots back when the parameter dies - with one
    # shared list the second registry's indices were released into the first one and freed a slot that belonged to a live parameter there (round-5 ADVICE)
    reg2 = T.Fp8Sites(torch.device("cpu"))
    live_before, free_before = dict(reg.live), list(reg.free)
    c = torch.nn.Parameter(torch.zeros(2))
    s_first = reg.site(c, "x")
    s_second = [reg2.site(c, "x"), reg2.site(c, "dy"), reg2.site(c, "w")]
    assert s_first.owner is reg and all(st.owner is reg2 for st in s_second) and len(reg2.live) == 3
    del c, s_first, s_second
    gc.collect()
    assert not reg2.live and sorted(reg2.free) == [0, 1, 2]
    assert reg.live.keys() == live_before.keys() and sorted(reg.free) == sorted(free_before), "the other registry's release must not touch this one"


def test_bench_gpus_n_without_a_launcher_launches_its_own_ranks(tmp_path: Path) -> None:
    """``python bench.py --gpus 2`` with no rank environment (the driver's command form) must not die on a launch convention: it re-executes itself under
    ``torch.distributed.run`` with two ranks.  Here (no GPU) a rank stops at the "needs an MI355X" check and says which rank of how many it is - which proves that
    the ranks were started with WORLD_SIZE=2 and reached ``main()`` past the WORLD_SIZE test.  (The launcher ends the other rank as soon as the first one fails, so
    only one of the two messages is guaranteed.)"""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=280, check=False,
                         cwd=str(tmp_path), env=env)
    assert out.returncode != 0
    assert "WORLD_SIZE=1" not in out.stderr and "launch with torch.distributed.run" not in out.stderr, out.stderr[-2000:]
    assert "bench.py needs an MI355X" in out.stderr and " of 2)" in out.stderr, out.stderr[-2000:]
