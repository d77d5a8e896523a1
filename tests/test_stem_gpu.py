"""Fused per-voxel halves of the conv stem's MaskedConvBlock (csrc/stem.hip) against plain PyTorch fp32 statements of the same arithmetic on the same bf16-rounded
operands (reference: cinema/conv.py:405-413; ConvLayerNorm eps 1e-6, ConvMlp = fc1 -> exact GELU -> fc2).  Through the C-ABI (cinema_amd.hip).  Needs an MI355X."""

from __future__ import annotations

import pytest
import torch
import torch.nn.functional as F  # noqa: N812

pytestmark = pytest.mark.gpu

from cinema_amd import hip as K  # noqa: E402

DEV = "cuda"
EPS = 1e-6


def rnd(*shape, scale=1.0, dtype=torch.float32, seed=0):  # noqa: ANN001, ANN002, ANN201
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20))


def bf(t: torch.Tensor) -> torch.Tensor:
    """Round to bf16 and back (the value an MFMA operand carries), differentiable as the identity."""
    return t + (t.to(torch.bfloat16).float() - t).detach()


def block_params(c: int, seed: int) -> dict:
    h = 4 * c
    p = {
        "g1": 1 + rnd(c, scale=0.2, seed=seed + 1), "b1n": rnd(c, scale=0.2, seed=seed + 2),
        "w1": rnd(c, c, scale=c ** -0.5, seed=seed + 3), "b1": rnd(c, scale=0.1, seed=seed + 4),
        "w2": rnd(c, c, scale=c ** -0.5, seed=seed + 5), "b2": rnd(c, scale=0.1, seed=seed + 6),
        "g2": 1 + rnd(c, scale=0.2, seed=seed + 7), "b2n": rnd(c, scale=0.2, seed=seed + 8),
        "wf1": rnd(h, c, scale=c ** -0.5, seed=seed + 9), "bf1": rnd(h, scale=0.1, seed=seed + 10),
        "wf2": rnd(c, h, scale=h ** -0.5, seed=seed + 11), "bf2": rnd(c, scale=0.1, seed=seed + 12),
    }
    for k in ("w1", "w2", "wf1", "wf2"):
        p[k + "_16"] = p[k].to(torch.bfloat16).contiguous()
    return p


SHAPES = [(64, 1000), (64, 4096), (128, 777), (128, 2304), (64, 9), (128, 20000)]


@pytest.mark.parametrize(("c", "rows"), SHAPES)
def test_stem_ln_linear_forward_and_backward(c: int, rows: int) -> None:
    p = block_params(c, 10)
    x = rnd(rows, c, scale=1.5, seed=1) + 0.3
    xn, h = K.stem_ln_linear(x, p["g1"], p["b1n"], EPS, p["w1_16"], p["b1"])
    xr = x.clone().requires_grad_(True)
    g1, b1n = p["g1"].clone().requires_grad_(True), p["b1n"].clone().requires_grad_(True)
    xn_ref = F.layer_norm(xr, (c,), g1, b1n, EPS)
    h_ref = bf(xn_ref) @ p["w1_16"].float().t() + p["b1"]
    assert rel_l2(xn, xn_ref) <= 4e-3, rel_l2(xn, xn_ref)      # one bf16 rounding
    assert rel_l2(h, h_ref) <= 4e-3, rel_l2(h, h_ref)
    # backward: dx = dres + LN'(x)(dh W)
    dh = rnd(rows, c, seed=2, dtype=torch.bfloat16)
    dres = rnd(rows, c, seed=3)
    h_ref.backward(dh.float())
    dx, (part, n_part) = K.stem_ln_linear_bwd(dh, x, dres, p["g1"], EPS, p["w1_16"])
    dx_ref = xr.grad + dres
    assert rel_l2(dx, dx_ref) <= 6e-3, rel_l2(dx, dx_ref)      # the data gradient dh W is rounded to bf16 accumulators' inputs only: fp32 chain, bf16 operands
    dg = part[:n_part, :c].sum(0)
    db = part[:n_part, c:].sum(0)
    assert rel_l2(dg, g1.grad) <= 6e-3 and rel_l2(db, b1n.grad) <= 6e-3, (rel_l2(dg, g1.grad), rel_l2(db, b1n.grad))
    dx0, _ = K.stem_ln_linear_bwd(dh, x, None, p["g1"], EPS, p["w1_16"])
    assert rel_l2(dx0, xr.grad) <= 6e-3


def mlp_reference(p: dict, d: torch.Tensor, x: torch.Tensor):  # noqa: ANN201
    """fp32 statement of the half block with the operand roundings of the kernel: bf16 d, LN output, GELU output; fp32 residual stream."""
    c = x.shape[1]
    x1 = x + d @ p["w2_16"].float().t() + p["b2"]
    xn2 = F.layer_norm(x1, (c,), p["g2"], p["b2n"], EPS)
    z = bf(xn2) @ p["wf1_16"].float().t() + p["bf1"]
    a = F.gelu(z)
    x2 = x1 + bf(a) @ p["wf2_16"].float().t() + p["bf2"]
    return x1, xn2, z, a, x2


@pytest.mark.parametrize(("c", "rows"), SHAPES)
def test_stem_mlp_forward(c: int, rows: int) -> None:
    p = block_params(c, 20)
    x = rnd(rows, c, scale=1.2, seed=4)
    d = rnd(rows, c, seed=5, dtype=torch.bfloat16)
    x1, x2 = K.stem_mlp_fwd(d, x, p["w2_16"], p["b2"], p["g2"], p["b2n"], EPS, p["wf1_16"], p["bf1"], p["wf2_16"], p["bf2"])
    x1_ref, _, _, _, x2_ref = mlp_reference(p, d.float(), x)
    assert rel_l2(x1, x1_ref) <= 1e-5, rel_l2(x1, x1_ref)       # fp32 accumulation order only
    assert rel_l2(x2, x2_ref) <= 2e-3, rel_l2(x2, x2_ref)       # bf16 rounding boundaries of the hidden layer may fall differently
    _, x2b = K.stem_mlp_fwd(d, x, p["w2_16"], p["b2"], p["g2"], p["b2n"], EPS, p["wf1_16"], p["bf1"], p["wf2_16"], p["bf2"], want_x1=False)
    assert torch.equal(x2, x2b)


@pytest.mark.parametrize(("c", "rows"), SHAPES)
def test_stem_mlp_backward(c: int, rows: int) -> None:
    p = block_params(c, 30)
    x = rnd(rows, c, scale=1.2, seed=6)
    d = rnd(rows, c, seed=7, dtype=torch.bfloat16)
    g2 = rnd(rows, c, seed=8)
    pr = {k: (v.clone().requires_grad_(True) if k in ("g2", "b2n") else v) for k, v in p.items()}
    dr = d.float().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    x1_ref, xn2_ref, z_ref, a_ref, x2_ref = mlp_reference(pr, dr, xr)
    z_ref.retain_grad()
    x1_ref.retain_grad()
    x2_ref.backward(g2)
    o = K.stem_mlp_bwd(g2, x1_ref.detach().contiguous(), p["w2_16"], p["g2"], p["b2n"], EPS, p["wf1_16"], p["bf1"], p["wf2_16"])
    assert rel_l2(o["xn2"], xn2_ref) <= 4e-3 and rel_l2(o["g2_16"], g2) <= 4e-3
    assert rel_l2(o["a"], a_ref) <= 6e-3, rel_l2(o["a"], a_ref)
    assert rel_l2(o["dz"], z_ref.grad) <= 1e-2, rel_l2(o["dz"], z_ref.grad)      # the reference keeps dL/da in fp32, the kernel's MFMA takes bf16 g2
    assert rel_l2(o["dx1"], x1_ref.grad) <= 6e-3, rel_l2(o["dx1"], x1_ref.grad)  # = dL/dx (the residual path is the identity)
    assert rel_l2(o["dx1_16"], x1_ref.grad) <= 8e-3
    assert rel_l2(o["dd"], dr.grad) <= 1e-2, rel_l2(o["dd"], dr.grad)
    part, n_part = o["partials"]
    dg, db = part[:n_part, :c].sum(0), part[:n_part, c:].sum(0)
    assert rel_l2(dg, pr["g2"].grad) <= 1e-2 and rel_l2(db, pr["b2n"].grad) <= 1e-2, (rel_l2(dg, pr["g2"].grad), rel_l2(db, pr["b2n"].grad))


@pytest.mark.parametrize("c", [64, 128])
@pytest.mark.parametrize("rows", [100, 4096, 36864 + 7])
def test_stem_wgrad_matches_fp32_products_and_is_deterministic(c: int, rows: int) -> None:
    h = 4 * c
    g16, a = rnd(rows, c, seed=1, dtype=torch.bfloat16), rnd(rows, h, seed=2, dtype=torch.bfloat16)
    dz, xn2 = rnd(rows, h, seed=3, dtype=torch.bfloat16), rnd(rows, c, seed=4, dtype=torch.bfloat16)
    dx1, d = rnd(rows, c, seed=5, dtype=torch.bfloat16), rnd(rows, c, seed=6, dtype=torch.bfloat16)
    dh, xn = rnd(rows, c, seed=7, dtype=torch.bfloat16), rnd(rows, c, seed=8, dtype=torch.bfloat16)

    def run() -> list:
        outs = [(torch.ones(c, h, device=DEV), torch.ones(c, device=DEV)), (torch.zeros(h, c, device=DEV), torch.zeros(h, device=DEV)),
                (torch.zeros(c, c, device=DEV), torch.zeros(c, device=DEV)), (torch.zeros(c, c, device=DEV), None)]
        K.stem_wgrad([(g16, a, *outs[0]), (dz, xn2, *outs[1]), (dx1, d, *outs[2]), (dh, xn, *outs[3])])
        return outs

    outs = run()
    pairs = [(g16, a), (dz, xn2), (dx1, d), (dh, xn)]
    for i, ((dy, x), (dw, db)) in enumerate(zip(pairs, outs)):
        ref = dy.float().t() @ x.float() + (1.0 if i == 0 else 0.0)  # (accumulated into what was there)
        assert rel_l2(dw, ref) <= 1e-5, (i, rel_l2(dw, ref))
        if db is not None:
            assert rel_l2(db, dy.float().sum(0) + (1.0 if i == 0 else 0.0)) <= 1e-5, i
    again = run()
    for (dw, db), (dw2, db2) in zip(outs, again):
        assert torch.equal(dw, dw2) and (db is None or torch.equal(db, db2))  # ordered slab reduce: bit-identical run to run
