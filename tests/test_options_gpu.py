"""Transformer-block options that no shipped CineMA config sets but the reference constructors accept (``cinema/vit.py:446-609``, ``vit.py:332-337``), against vectors
written by the imported reference (``oracle/make_golden_block_options.py`` -> ``tests/golden/block_options.safetensors``): timm LayerScale (``init_values``), ``qk_norm``,
``PatchEmbed(dynamic_img_pad=True)``; and ``proj_drop`` (no reference vector can pin a random mask: structure and statistics)."""
from __future__ import annotations

import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu

from cinema_amd import hip as K  # noqa: E402
from cinema_amd.vit import Block, LayerScale, Mlp, PatchEmbed  # noqa: E402
from conftest import load_golden  # noqa: E402

DEV = "cuda"
CASES = {"layerscale_hd16": (64, 4, dict(init_values=0.1)), "layerscale_hd64_cross": (128, 2, dict(init_values=0.5)), "qknorm_hd16": (64, 4, dict(qk_norm=True)),
         "qknorm_hd64_cross": (128, 2, dict(qk_norm=True)), "qknorm_layerscale_hd32": (128, 4, dict(qk_norm=True, init_values=0.2))}


def make_block(dim: int, heads: int, **kw) -> Block:  # noqa: ANN003
    return Block(dim=dim, n_heads=heads, mlp_ratio=4, norm_layer=nn.LayerNorm, norm_eps=1e-6, drop_path=0.0, qkv_bias=True, rotary=False, act_layer=nn.GELU, mlp_layer=Mlp, **kw)


def rel(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.float().cpu() - b).norm() / b.norm().clamp_min(1e-12))


@pytest.mark.parametrize("name", list(CASES))
def test_block_options_vs_reference_vectors(name: str) -> None:
    """Output within 2 % rel-L2 of the reference's (bf16 GEMMs against fp32), every parameter gradient within 3 % (matrices) / 5 % (vectors: LayerNorm, bias, LayerScale gamma,
    q / k norm), input gradients within 3 %."""
    g = {k[len(name) + 1:]: v for k, v in load_golden("block_options.safetensors").items() if k.startswith(name + "/")}
    dim, heads, kw = CASES[name]
    blk = make_block(dim, heads, **kw)
    sd = {k[len("param/"):]: v for k, v in g.items() if k.startswith("param/")}
    assert set(sd) == set(blk.state_dict()), (sorted(set(sd) ^ set(blk.state_dict())))  # the reference's parameter names (ls1.gamma, attn.q_norm.weight, ...)
    blk.load_state_dict(sd)
    blk.to(DEV).eval()
    q = g["q"].to(DEV).requires_grad_(True)
    k = g["k"].to(DEV).requires_grad_(True) if "k" in g else None
    out = blk(q, k)
    assert rel(out, g["out"]) <= 2e-2, rel(out, g["out"])
    (out * g["w"].to(DEV)).sum().backward()
    worst = {"matrix": ("", 0.0), "vector": ("", 0.0)}
    gnorm = max(float(v.norm()) for kk, v in g.items() if kk.startswith("grad/"))
    for n, p in blk.named_parameters():
        if float(g[f"grad/{n}"].norm()) < 1e-4 * gnorm:
            # k_norm.bias: a constant added to every key of a head shifts each query's logits by one value, which the softmax ignores - the true gradient is ZERO and
            # the reference's value is rounding noise (1e-8): this path's value must be noise as well, measured against the largest gradient
            assert float(p.grad.float().norm()) <= 2e-3 * gnorm, (n, float(p.grad.float().norm()), gnorm)
            continue
        r = rel(p.grad, g[f"grad/{n}"])
        kind = "matrix" if p.dim() > 1 else "vector"
        if r > worst[kind][1]:
            worst[kind] = (n, r)
    print(name, "out", rel(out, g["out"]), "worst", worst, "dq", rel(q.grad, g["dq"]), "dk", None if k is None else rel(k.grad, g["dk"]))
    assert worst["matrix"][1] <= 3e-2 and worst["vector"][1] <= 5e-2, worst
    assert rel(q.grad, g["dq"]) <= 3e-2
    if k is not None:
        assert rel(k.grad, g["dk"]) <= 3e-2


def test_patch_embed_dynamic_pad_vs_reference_vectors() -> None:
    g = load_golden("block_options.safetensors")
    for name, size, patch in (("pad2d", (30, 30), (4, 4)), ("pad2d_aniso", (10, 14), (4, 8)), ("pad3d", (9, 9, 9), (4, 4, 4))):
        pe = PatchEmbed(image_size=size, patch_size=patch, in_chans=2, embed_dim=32, dynamic_img_pad=True)
        pe.proj.weight.data.copy_(g[f"{name}/weight"])
        pe.proj.bias.data.copy_(g[f"{name}/bias"])
        pe.to(DEV)
        y = pe(g[f"{name}/x"].to(DEV))
        assert y.shape == g[f"{name}/y"].shape and rel(y, g[f"{name}/y"]) <= 1e-2, (name, rel(y, g[f"{name}/y"]))
    # the reference hands its per-axis pad pairs to F.pad in reverse: where the amounts disagree its own patchify raises, and so does this one
    assert "pad3d_mixed/raises" in g
    pe = PatchEmbed(image_size=(10, 13, 7), patch_size=(4, 4, 2), in_chans=2, embed_dim=32, dynamic_img_pad=True).to(DEV)
    with pytest.raises(ValueError, match="divi"):
        pe(torch.randn(2, 2, 10, 13, 7, device=DEV))


def test_mul_rows_and_layerscale_module() -> None:
    torch.manual_seed(0)
    a, gam, b = torch.randn(300, 96, device=DEV), torch.randn(96, device=DEV), torch.randn(300, 96, device=DEV).to(torch.bfloat16)
    assert torch.allclose(K.mul_rows(a, gam), a * gam, atol=1e-6)
    assert torch.allclose(K.mul_rows(a.to(torch.bfloat16), b, torch.bfloat16).float(), (a.to(torch.bfloat16).float() * b.float()).to(torch.bfloat16).float(), atol=1e-2)
    ls = LayerScale(96, init_values=0.3).to(DEV)
    x = torch.randn(4, 75, 96, device=DEV, requires_grad=True)
    y = ls(x)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    assert torch.allclose(y, x * ls.gamma, atol=1e-6)
    assert torch.allclose(x.grad, w * ls.gamma, atol=1e-6) and torch.allclose(ls.gamma.grad, (w * x).sum((0, 1)), rtol=1e-4, atol=1e-3)


def test_proj_drop_is_identity_in_eval_and_a_scaled_mask_in_training() -> None:
    """``proj_drop`` = nn.Dropout behind the attention projection and twice inside timm's Mlp (``vit.py:483-484,570-575``).  Eval mode: the block equals the same block
    without dropout.  Training mode: new masks per forward; gradients finite; with the MLP reduced to a pass-through probe (fc2 = 0, attention projection = 0 except its
    bias) the branch output is bias * keep / (1 - p): every element is 0 or bias / (1 - p) and the kept fraction is 1 - p within 3 sigma."""
    torch.manual_seed(1)
    p = 0.25
    blk, ref = make_block(128, 4, proj_drop=p).to(DEV), make_block(128, 4).to(DEV)
    ref.load_state_dict(blk.state_dict())
    q = torch.randn(2, 48, 128, device=DEV)
    blk.eval(), ref.eval()
    assert torch.equal(blk(q), ref(q))
    blk.train()
    y1, y2 = blk(q), blk(q)
    assert not torch.equal(y1, y2)
    qg = q.clone().requires_grad_(True)
    blk(qg).sum().backward()
    assert torch.isfinite(qg.grad).all() and all(torch.isfinite(t.grad).all() for t in blk.parameters())
    with torch.no_grad():
        blk.attn.proj.weight.zero_()
        blk.attn.proj.bias.fill_(2.0)
        blk.mlp.fc2.weight.zero_()
        blk.mlp.fc2.bias.zero_()
    out = blk(q) - q  # = dropout(bias): 0 or 2 / (1 - p)
    kept = (out - 2.0 / (1 - p)).abs() < 2e-2
    assert bool((kept | (out.abs() < 1e-6)).all())
    n = out.numel()
    assert abs(float(kept.float().mean()) - (1 - p)) <= 3 * (p * (1 - p) / n) ** 0.5
