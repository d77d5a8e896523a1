"""Kernel-level parity: every C-ABI entry point (through cinema_amd.hip) against a plain PyTorch fp32 statement of
the same op on the same (bf16-rounded) inputs.  Tolerances are written next to each check.  Needs an MI355X."""

from __future__ import annotations

import math

import pytest
import torch
import torch.nn.functional as F  # noqa: N812

pytestmark = pytest.mark.gpu

import cinema_oracle as O  # noqa: E402
from cinema_amd import hip as K  # noqa: E402

DEV = "cuda"


def rnd(*shape, scale=1.0, dtype=torch.bfloat16, seed=None):  # noqa: ANN001, ANN002, ANN201
    g = torch.Generator(device="cpu")
    g.manual_seed(seed if seed is not None else (hash(shape) % 10007))
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def close(a: torch.Tensor, b: torch.Tensor, rtol: float, atol: float, what: str = "") -> None:
    a, b = a.float(), b.float()
    err = (a - b).abs()
    bound = atol + rtol * b.abs()
    bad = err > bound
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} mismatches, max abs err {float(err.max()):.4g}, max ref {float(b.abs().max()):.4g}"


def test_library_info() -> None:
    info = K.info()
    assert info["abi_version"] == 1 and info["wave_size"] == 64 and info["n_cus"] > 0


# ------------------------------------------------------------------------------------------------ GEMM
GEMM_SHAPES = [(300, 256, 192), (1000, 768, 768), (129, 16, 16), (2053, 512, 2048), (70, 24, 40), (128, 128, 64), (4100, 64, 256)]


@pytest.mark.parametrize(("m", "n", "k"), GEMM_SHAPES)
@pytest.mark.parametrize("generic", [False, True])
def test_gemm_forward_layout(m: int, n: int, k: int, generic: bool) -> None:
    a, w = rnd(m, k, seed=1), rnd(n, k, seed=2)
    bias = rnd(n, dtype=torch.float32, seed=3)
    ref = a.float() @ w.float().t() + bias
    out = K.gemm(a, w, bias=bias, out_dtype=torch.float32, force_generic=generic)
    close(out, ref, 2e-4, 2e-3, "fwd f32")  # fp32 accumulation-order noise only
    out16 = K.gemm(a, w, bias=bias, force_generic=generic)
    close(out16, ref, 1e-2, 2e-2, "fwd bf16")  # one bf16 rounding of the output


@pytest.mark.parametrize(("m", "n", "k"), [(300, 256, 192), (1000, 768, 3072), (129, 16, 64), (685, 768, 768)])
@pytest.mark.parametrize("generic", [False, True])
def test_gemm_dgrad_and_wgrad_layouts(m: int, n: int, k: int, generic: bool) -> None:
    """Y = X W^T with X[m,k], W[n,k]:  dX = dY W  (A k-major, B stored [red][out]);  dW = dY^T X (both reduction-strided)."""
    x, w, dy = rnd(m, k, seed=4), rnd(n, k, seed=5), rnd(m, n, seed=6)
    dx_ref = dy.float() @ w.float()
    dx = K.gemm(dy, w, a_kmajor=True, b_kmajor=False, out_dtype=torch.float32, force_generic=generic)
    close(dx, dx_ref, 2e-4, 5e-3, "dgrad")
    dw_ref = dy.float().t() @ x.float()
    dw = K.gemm(dy, x, a_kmajor=False, b_kmajor=False, out_dtype=torch.float32, force_generic=generic)
    close(dw, dw_ref, 2e-4, 1e-2, "wgrad")
    acc = torch.ones(n, k, dtype=torch.float32, device=DEV)
    K.gemm(dy, x, a_kmajor=False, b_kmajor=False, out=acc, accumulate=True, split_k=5, force_generic=generic)
    close(acc, dw_ref + 1.0, 2e-4, 1e-2, "wgrad split-k accumulate")
    # fused bias gradient: a_rowsum[n] += sum_m dy[m, n], accumulated on top of existing content
    rs = torch.full((n,), 0.5, dtype=torch.float32, device=DEV)
    acc2 = torch.zeros(n, k, dtype=torch.float32, device=DEV)
    K.gemm(dy, x, a_kmajor=False, b_kmajor=False, out=acc2, accumulate=True, split_k=3, force_generic=generic, a_rowsum=rs)
    close(acc2, dw_ref, 2e-4, 1e-2, "wgrad with fused row sums")
    close(rs, dy.float().sum(0) + 0.5, 1e-3, 2e-2, "fused bias gradient")


@pytest.mark.parametrize("generic", [False, True])
def test_gemm_epilogues(generic: bool) -> None:
    m, n, k = 517, 256, 128
    a, w = rnd(m, k, seed=7), rnd(n, k, seed=8, scale=0.2)
    bias = rnd(n, dtype=torch.float32, seed=9)
    pre = a.float() @ w.float().t() + bias
    aux = torch.empty(m, n, dtype=torch.bfloat16, device=DEV)
    out = K.gemm(a, w, bias=bias, act=1, aux_out=aux, force_generic=generic)
    close(aux, pre, 1e-2, 2e-2, "aux pre-activation")
    close(out, F.gelu(pre), 1e-2, 2e-2, "gelu epilogue")
    res = rnd(m, n, dtype=torch.float32, seed=10)
    out = K.gemm(a, w, bias=bias, residual=res, out_dtype=torch.float32, force_generic=generic)
    close(out, pre + res, 2e-4, 2e-3, "fp32 residual")
    out_inplace = res.clone()
    K.gemm(a, w, bias=bias, residual=out_inplace, out=out_inplace, force_generic=generic)
    close(out_inplace, pre + res, 2e-4, 2e-3, "in-place residual")
    res16 = rnd(m, n, seed=11)
    out = K.gemm(a, w, residual=res16, out_dtype=torch.float32, alpha=0.5, force_generic=generic)
    close(out, 0.5 * (a.float() @ w.float().t()) + res16.float(), 2e-4, 2e-3, "bf16 residual + alpha")
    mask = (torch.arange(m, device=DEV) % 3 != 0).to(torch.uint8)
    out = K.gemm(a, w, bias=bias, row_mask=mask, out_dtype=torch.float32, force_generic=generic)
    close(out, pre * mask[:, None].float(), 2e-4, 2e-3, "row mask")
    h = rnd(m, n, seed=12)
    hx = h.float().requires_grad_(True)
    F.gelu(hx).backward(torch.ones_like(hx))
    out = K.gemm(a, w, gelu_in=h, out_dtype=torch.float32, force_generic=generic)
    close(out, (a.float() @ w.float().t()) * hx.grad, 5e-4, 5e-3, "gelu' epilogue")


@pytest.mark.parametrize("k", [128, 1024])  # BK = 32 kernel (K <= 512) and BK = 64 kernel
def test_gemm_bf16_epilogue_classes_on_both_tile_depths(k: int) -> None:
    """The compile-time epilogue classes with a bf16 output (plain, GELU + pre-activation, dY x GELU'(pre-activation) on the data-gradient
    layout) on both kernels; tolerance = one bf16 rounding of the output."""
    m, n = 1300, 384
    a, w = rnd(m, k, seed=21), rnd(n, k, seed=22, scale=0.1)
    bias = rnd(n, dtype=torch.float32, seed=23)
    pre = a.float() @ w.float().t() + bias
    aux = torch.empty(m, n, dtype=torch.bfloat16, device=DEV)
    out = K.gemm(a, w, bias=bias, act=1, aux_out=aux)
    close(aux, pre, 1e-2, 2e-2, "pre-activation")
    close(out, F.gelu(pre), 1e-2, 2e-2, "gelu")
    close(K.gemm(a, w, bias=bias), pre, 1e-2, 2e-2, "plain bf16")
    dy, w2 = rnd(m, k, seed=24), rnd(k, n, seed=25, scale=0.1)  # dh[m, n] = (dy[m, k] @ W2[k, n]) * gelu'(pre[m, n])
    hx = aux.float().requires_grad_(True)
    F.gelu(hx).backward(torch.ones_like(hx))
    dh = K.gemm(dy, w2, a_kmajor=True, b_kmajor=False, gelu_in=aux)
    assert dh.dtype == torch.bfloat16
    close(dh, (dy.float() @ w2.float()) * hx.grad, 1e-2, 2e-2, "gelu' on the data-gradient layout")
    # stored-derivative form (gelu_deriv): the forward epilogue writes GELU'(pre-activation) from the same erf terms, the data gradient multiplies it in
    dv = torch.empty(m, n, dtype=torch.bfloat16, device=DEV)
    assert torch.equal(K.gemm(a, w, bias=bias, act=1, aux_out=dv, gelu_deriv=True), out)
    px = pre.clone().requires_grad_(True)
    F.gelu(px).backward(torch.ones_like(px))
    close(dv, px.grad, 1e-2, 1e-2, "stored GELU derivative")
    dh2 = K.gemm(dy, w2, a_kmajor=True, b_kmajor=False, gelu_in=dv, gelu_deriv=True)
    close(dh2, (dy.float() @ w2.float()) * dv.float(), 1e-2, 2e-2, "data gradient x stored derivative")
    # the derivative as an 8-bit code (uint8 tensor: affine map of [-0.13, 1.13] onto 0..255, csrc/common.cuh): half the bytes; error <= half a step
    # (0.00247) against the exact derivative, the data gradient multiplies the DECODED value (bit-for-bit what the decoder formula gives)
    dv8 = torch.empty(m, n, dtype=torch.uint8, device=DEV)
    assert torch.equal(K.gemm(a, w, bias=bias, act=1, aux_out=dv8, gelu_deriv=True), out)
    dec = dv8.float() * (1.26 / 255.0) - 0.13
    assert float((dec - px.grad).abs().max()) <= 0.00247 + 2e-4, float((dec - px.grad).abs().max())  # (+ the erf approximation, 1.5e-7, and the products' fp32 rounding)
    assert float(px.grad.min()) >= -0.13 and float(px.grad.max()) <= 1.13
    dh8 = K.gemm(dy, w2, a_kmajor=True, b_kmajor=False, gelu_in=dv8, gelu_deriv=True)
    close(dh8, (dy.float() @ w2.float()) * dec, 1e-2, 2e-2, "data gradient x 8-bit derivative code")
    g8 = torch.empty(m, n, dtype=torch.uint8, device=DEV)
    K.gemm(a, w, bias=bias, act=1, aux_out=g8, gelu_deriv=True, force_generic=True)
    assert int((g8.int() - dv8.int()).abs().max()) <= 1  # the generic kernel's code: at most one step apart (different summation order of the pre-activation)
    close(K.gemm(dy, w2, a_kmajor=True, b_kmajor=False, gelu_in=dv8, gelu_deriv=True, force_generic=True), (dy.float() @ w2.float()) * dec, 1e-2, 2e-2, "generic x 8-bit code")
    with pytest.raises(K.HipLibraryError):
        K.gemm(dy, w2, a_kmajor=True, b_kmajor=False, gelu_in=dv8)  # a uint8 tensor is never a pre-activation
    for sched in (0, 1):  # the same two epilogues on the persistent 256x256 kernel
        dv3 = torch.empty_like(dv)
        d8p = torch.empty_like(dv8)
        K.gemm(a, w, bias=bias, act=1, aux_out=d8p, gelu_deriv=True, p256=sched, split_k=0)
        assert int((d8p.int() - dv8.int()).abs().max()) <= 1
        close(K.gemm(dy, w2, a_kmajor=True, b_kmajor=False, gelu_in=dv8, gelu_deriv=True, p256=sched, split_k=0), (dy.float() @ w2.float()) * dec, 1e-2, 2e-2, "p256 x 8-bit code")
        close(K.gemm(a, w, bias=bias, act=1, aux_out=dv3, gelu_deriv=True, p256=sched, split_k=0), F.gelu(pre), 1e-2, 2e-2, "p256 gelu")
        close(dv3, px.grad, 1e-2, 1e-2, "p256 stored derivative")
        close(K.gemm(dy, w2, a_kmajor=True, b_kmajor=False, gelu_in=dv, gelu_deriv=True, p256=sched, split_k=0), (dy.float() @ w2.float()) * dv.float(), 1e-2, 2e-2,
              "p256 x stored derivative")


@pytest.mark.parametrize(("m", "n", "k"), [(10960, 768, 3072), (8300, 512, 2048), (10960, 768, 2304)])
def test_gemm_split_tail_full_size(m: int, n: int, k: int) -> None:
    """BASELINE config-2 shapes whose tile count just exceeds the 512 workgroup slots (516 / 260 tiles): the left-over tiles
    run as k-slices + fix-up launch.  Same fused epilogues, same tolerances as the single-launch path."""
    a, w = rnd(m, k, scale=0.5, seed=11), rnd(n, k, scale=0.05, seed=12)
    bias = rnd(n, dtype=torch.float32, seed=13)
    res = rnd(m, n, dtype=torch.float32, seed=14)
    ref = a.float() @ w.float().t() + bias
    out = K.gemm(a, w, bias=bias, residual=res, out_dtype=torch.float32)
    close(out, ref + res, 2e-4, 2e-3, "tail f32 + residual")
    pre = torch.empty(m, n, dtype=torch.bfloat16, device=DEV)
    act = K.gemm(a, w, bias=bias, act=1, aux_out=pre)
    close(pre, ref, 1e-2, 2e-2, "tail pre-activation")
    close(act, F.gelu(ref), 1e-2, 2e-2, "tail gelu")
    wt = w.t().contiguous()  # dgrad layout: B stored [k][n]
    out2 = K.gemm(a, wt, b_kmajor=False, out_dtype=torch.float32)
    close(out2, ref - bias, 2e-4, 2e-3, "tail dgrad layout")
    # both forms of the tail: the k-slices finished INSIDE the launch (default: every slice sums and finishes its share of the tile, slices summed in slice
    # order) and the fix-up launch: bit-identical to each other, run to run, and the counters / the error word are left zero
    prev = K.TAIL_IN_LAUNCH
    try:
        K.TAIL_IN_LAUNCH = False
        out_f, act_f, out2_f = K.gemm(a, w, bias=bias, residual=res, out_dtype=torch.float32), K.gemm(a, w, bias=bias, act=1), K.gemm(a, wt, b_kmajor=False, out_dtype=torch.float32)
        K.TAIL_IN_LAUNCH = True
        for _ in range(3):
            assert torch.equal(K.gemm(a, w, bias=bias, residual=res, out_dtype=torch.float32), out_f)
            assert torch.equal(K.gemm(a, w, bias=bias, act=1), act_f)
            assert torch.equal(K.gemm(a, wt, b_kmajor=False, out_dtype=torch.float32), out2_f)
    finally:
        K.TAIL_IN_LAUNCH = prev
    torch.cuda.synchronize()
    assert int(K._tail_counters(a.device).abs().sum()) == 0


def test_gemm_wgrad_grouped_matches_single_launches() -> None:
    """cinema_gemm_bf16_grouped: several weight gradients in one launch (whole-K tiles, no split-K slabs) == the per-GEMM launches
    (deterministic split-K) and the fp32 reference; fp32 accumulation order differs, tolerance 1e-3 relative to the largest element."""
    rows = 1000
    shapes = [(256, 384), (128, 128), (384, 256), (64, 512)]
    probs, single, ref = [], [], []
    for i, (n, k) in enumerate(shapes):
        dy = rnd(rows, n, scale=0.5, seed=80 + i)
        x = rnd(rows, k, scale=0.5, seed=90 + i)
        base = rnd(n, k, dtype=torch.float32, seed=100 + i)
        bsum = rnd(n, dtype=torch.float32, seed=110 + i)
        dst, b1 = base.clone(), bsum.clone()
        probs.append((dy, x, dst, b1))
        d2, b2 = base.clone(), bsum.clone()
        K.gemm(dy, x, a_kmajor=False, b_kmajor=False, out=d2, accumulate=True, split_k=3, a_rowsum=b2)
        single.append((d2, b2))
        ref.append((base + dy.float().t() @ x.float(), bsum + dy.float().sum(0)))
    K.gemm_wgrad_grouped(probs)
    for (dy, x, dst, b1), (d2, b2), (rd, rb) in zip(probs, single, ref):
        close(dst, rd, 0.0, 1e-3 * float(rd.abs().max()), "grouped dW vs fp32")
        close(dst, d2, 0.0, 1e-3 * float(rd.abs().max()), "grouped dW vs single launch")
        close(b1, rb, 0.0, 1e-3 * float(rb.abs().max()), "grouped bias gradient")
    with pytest.raises(K.HipLibraryError):  # forward-layout problems are not groupable
        K.gemm_wgrad_grouped([(rnd(64, 8), rnd(32, 8), torch.zeros(8, 8, device=DEV), None)])


@pytest.mark.parametrize("remainder", [0, 1])
@pytest.mark.parametrize("rows", [700, 5000, 10960])
def test_gemm_p256_grouped_wgrad_in_launch_reduction(rows: int, remainder: int, monkeypatch: pytest.MonkeyPatch) -> None:
    """cinema_gemm_bf16_p256, split schedule: the weight gradients of a block in one persistent launch, tiles cut into k-slices and finished by their
    last-arriving piece inside the launch.  Against fp32 torch (1e-3 of the largest element: accumulation order only), twice on the same workspace
    (the counters must come back to zero), ragged tile edges (n, k not multiples of 256) and a ragged last k-tile (rows % 64 != 0) included.
    ``remainder`` = 1: the split + remainder schedule (equal-length slices, the rests of several tiles on one workgroup: fan-in 3, pieces of two lengths)."""
    monkeypatch.setenv("CINEMA_P256_REMAINDER", str(remainder))
    shapes = [(512, 256), (256, 768), (384, 200), (72, 512)] if rows < 10000 else [(2304, 768), (768, 768), (3072, 768), (768, 3072)]  # an encoder block
    for rep in range(2):
        probs, ref = [], []
        for i, (n, k) in enumerate(shapes):
            dy = rnd(rows, n, scale=0.5, seed=80 + i + 7 * rep)
            x = rnd(rows, k, scale=0.5, seed=90 + i + 7 * rep)
            base = rnd(n, k, dtype=torch.float32, seed=100 + i)
            bsum = rnd(n, dtype=torch.float32, seed=110 + i)
            probs.append((dy, x, base.clone(), bsum.clone()))
            ref.append((base + dy.float().t() @ x.float(), bsum + dy.float().sum(0)))
        K.gemm_wgrad_grouped(probs, p256=True)
        for (dy, x, dst, b1), (rd, rb) in zip(probs, ref):
            close(dst, rd, 0.0, 1e-3 * float(rd.abs().max()), f"p256 grouped dW rep {rep}")
            close(b1, rb, 0.0, 1e-3 * float(rb.abs().max()), f"p256 grouped bias gradient rep {rep}")
    ws = K._p256_workspace(torch.device(DEV, torch.cuda.current_device()))
    assert int(ws[:16384].view(torch.int32).abs().sum()) == 0, "tile counters / error word not left at zero"


@pytest.mark.parametrize("rows", [200, 5000])
def test_gemm_p256_main_loop_forms_agree_bit_for_bit(rows: int, monkeypatch: pytest.MonkeyPatch) -> None:
    """CINEMA_P256_LOOP: form 2 (default: the reading wave issues the LDS-DMA of the phase three ahead) against form 1 (issued between the MFMAs) and form 0 (k-tile
    loop): the same pieces, the same summation order - identical bits on the weight-gradient, forward and data-gradient layouts; short pieces (fewer phases than
    the ring is deep: rows = 200 -> 7 phases cut into slices) exercise the counted-vmcnt tail of the loop."""
    shapes = [(512, 256), (384, 200), (72, 512)]
    x, w = rnd(rows, 520, seed=3), rnd(264, 520, scale=0.05, seed=4)
    dy2 = rnd(rows, 264, seed=5)
    outs = {}
    for form in (2, 1, 0):
        monkeypatch.setenv("CINEMA_P256_LOOP", str(form))
        probs = [(rnd(rows, n, scale=0.5, seed=80 + i), rnd(rows, k, scale=0.5, seed=90 + i), torch.zeros(n, k, device=DEV), torch.zeros(n, device=DEV)) for i, (n, k) in enumerate(shapes)]
        K.gemm_wgrad_grouped(probs, p256=True)
        y = torch.empty(rows, 264, dtype=torch.bfloat16, device=DEV)
        dx = torch.empty(rows, 520, dtype=torch.bfloat16, device=DEV)
        for sched in (0, 1):
            K.gemm(x, w, out=y, p256=sched)
            K.gemm(dy2, w, a_kmajor=True, b_kmajor=False, out=dx, p256=sched)
            outs.setdefault((form, sched), []).extend([y.clone(), dx.clone()])
        outs[(form, "w")] = [dst for _, _, dst, _ in probs]
    for key in ("w", 0, 1):
        for a, b in zip(outs[(2, key)], outs[(1, key)]):
            assert torch.equal(a, b), f"forms 1 and 2 differ ({key})"
    for a, b in zip(outs[(2, "w")], outs[(0, "w")]):  # form 0 walks k-tiles of 64: same slices only when the k extents line up - compare numerically
        close(a, b, 0.0, 1e-3 * float(b.abs().max()), "form 0 vs 2")


@pytest.mark.parametrize("schedule", [0, 1])
@pytest.mark.parametrize(("m", "n", "k"), [(1000, 768, 3072), (2053, 512, 2048), (685, 256, 768), (300, 264, 200)])
def test_gemm_p256_forward_and_dgrad_epilogues(schedule: int, m: int, n: int, k: int) -> None:
    """The persistent 256x256 kernel on the forward and data-gradient layouts with every fused epilogue class, both schedules (balanced k-slices and
    stream ranges: tiles cut at arbitrary k-tiles, up to three pieces), against fp32 torch with the tolerances of the 128x128 kernel's tests."""
    a, w = rnd(m, k, seed=1), rnd(n, k, scale=0.05, seed=2)
    bias = rnd(n, dtype=torch.float32, seed=3)
    res = rnd(m, n, dtype=torch.float32, seed=7)
    ref = a.float() @ w.float().t() + bias
    scale = float(ref.abs().max())
    close(K.gemm(a, w, bias=bias, out_dtype=torch.float32, p256=schedule, split_k=0), ref, 2e-4, 2e-4 * scale, "p256 fwd f32")
    close(K.gemm(a, w, bias=bias, residual=res, out_dtype=torch.float32, p256=schedule, split_k=0), ref + res, 2e-4, 2e-4 * scale, "p256 fwd f32 + residual")
    close(K.gemm(a, w, bias=bias, p256=schedule, split_k=0), ref, 1e-2, 1e-2 * scale, "p256 fwd bf16")
    h = torch.empty(m, n, dtype=torch.bfloat16, device=DEV)
    act = K.gemm(a, w, bias=bias, act=1, aux_out=h, p256=schedule, split_k=0)
    close(h, ref, 1e-2, 1e-2 * scale, "p256 pre-activation copy")
    close(act, F.gelu(ref), 1e-2, 1e-2 * scale, "p256 gelu")
    dy = rnd(m, n, seed=6)
    wt = w  # dX = dY W with W stored [n (reduction)][k (out)]
    dref = dy.float() @ wt.float()
    dscale = float(dref.abs().max())
    close(K.gemm(dy, wt, a_kmajor=True, b_kmajor=False, out_dtype=torch.float32, p256=schedule, split_k=0), dref, 2e-4, 3e-4 * dscale, "p256 dgrad f32")
    pre = rnd(m, k, seed=8)
    gref = dref * (0.5 * (1 + torch.erf(pre.float() / math.sqrt(2))) + pre.float() * torch.exp(-0.5 * pre.float() ** 2) / math.sqrt(2 * math.pi))
    close(K.gemm(dy, wt, a_kmajor=True, b_kmajor=False, gelu_in=pre, p256=schedule, split_k=0), gref, 1e-2, 1e-2 * dscale, "p256 dgrad x gelu'")
    # whole-K tiles (split_k = 1): the plain persistent form, no partial slots touched
    close(K.gemm(a, w, bias=bias, out_dtype=torch.float32, p256=0, split_k=1), ref, 2e-4, 2e-4 * scale, "p256 whole-K")
    ws = K._p256_workspace(torch.device(DEV, torch.cuda.current_device()))
    assert int(ws[:16384].view(torch.int32).abs().sum()) == 0, "tile counters / error word not left at zero"


def test_gemm_strided_views_and_colsum() -> None:
    big = rnd(300, 3 * 256, seed=13)
    a = big[:, 256:512]  # column slice of a fused buffer
    w = rnd(128, 256, seed=14)
    close(K.gemm(a, w, out_dtype=torch.float32), a.float() @ w.float().t(), 2e-4, 2e-3, "lda > k")
    out = torch.zeros(768, dtype=torch.float32, device=DEV)
    K.colsum(big, out)
    close(out, big.float().sum(0), 1e-4, 1e-3, "colsum")
    # fp32 rows gathered through an index list (token-parameter gradients): vectorised kernel (n % 4 == 0) and scalar kernel (n = 6)
    for n in (512, 6):
        x32 = rnd(5000, n, dtype=torch.float32, seed=15)
        idx = torch.randperm(5000, generator=torch.Generator().manual_seed(4))[:3333].to(torch.int32).to(DEV)
        acc = torch.full((n,), 2.0, dtype=torch.float32, device=DEV)
        K.colsum(x32, acc, row_idx=idx)
        close(acc, x32[idx.long()].sum(0) + 2.0, 1e-4, 2e-3, f"gathered fp32 colsum n={n}")
    # narrow tall matrices (bias gradient of the 4-class segmentation head over millions of voxels): folded into 64-wide rows
    for n, dt in ((4, torch.bfloat16), (2, torch.float32), (1, torch.bfloat16)):
        tall = rnd(1 << 17, n, dtype=dt, seed=16)
        acc = torch.full((n,), -1.0, dtype=torch.float32, device=DEV)
        K.colsum(tall, acc)
        close(acc, tall.float().sum(0) - 1.0, 1e-3, 2e-2, f"narrow colsum n={n}")


# ------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("c", [16, 24, 64, 128, 512, 768, 1024])
@pytest.mark.parametrize("act", [0, 1])
@pytest.mark.parametrize("x_bf16", [False, True])
def test_layernorm(c: int, act: int, x_bf16: bool) -> None:
    rows = 333
    x = rnd(rows, c, dtype=torch.bfloat16 if x_bf16 else torch.float32, seed=20, scale=2.0) + 0.5
    gamma, beta = rnd(c, dtype=torch.float32, seed=21) + 1.0, rnd(c, dtype=torch.float32, seed=22)
    xr = x.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y_ref = F.layer_norm(xr, (c,), gr, br, 1e-6)
    if act:
        y_ref = F.gelu(y_ref)
    y16, y32, mean, rstd = K.layernorm_fwd(x, gamma, beta, 1e-6, act=act, want_bf16=True, want_f32=True)
    close(y32, y_ref, 1e-4, 1e-4, "ln fwd f32")
    close(y16, y_ref, 1e-2, 1e-2, "ln fwd bf16")
    dy = rnd(rows, c, seed=23)
    y_ref.backward(dy.float())
    resid = rnd(rows, c, dtype=torch.float32, seed=24)
    dgamma, dbeta = torch.zeros_like(gamma), torch.zeros_like(beta)
    dx32, dx16 = K.layernorm_bwd(dy, x, gamma, beta, mean, rstd, act=act, dx_residual=resid, want_f32=True, want_bf16=True, dgamma=dgamma,
                                 dbeta=dbeta)
    close(dx32, xr.grad + resid, 1e-3, 1e-3, "ln dx")
    close(dx16, xr.grad + resid, 1e-2, 2e-2, "ln dx bf16")
    close(dgamma, gr.grad, 1e-3, 2e-3, "ln dgamma")
    close(dbeta, br.grad, 1e-3, 2e-3, "ln dbeta")


# ------------------------------------------------------------------------------------------------ attention
def attn_ref(q, k, v, heads):  # noqa: ANN001, ANN201
    b, tq, c = q.shape
    hd = c // heads
    qh = q.float().reshape(b, tq, heads, hd).transpose(1, 2)
    kh = k.float().reshape(b, -1, heads, hd).transpose(1, 2)
    vh = v.float().reshape(b, -1, heads, hd).transpose(1, 2)
    w = torch.softmax(qh @ kh.transpose(-1, -2) * hd**-0.5, dim=-1)
    return (w @ vh).transpose(1, 2).reshape(b, tq, c)


def test_layernorm_deferred_param_gradients() -> None:
    """cinema_layernorm_bwd_deferred + cinema_ln_param_reduce_batched (one reduce for many LayerNorms) == the per-launch reduction."""
    items, direct = [], []
    for i, (rows, c) in enumerate([(5000, 512), (3000, 768), (9000, 64), (40, 128)]):  # the last one is too small for partials: added directly
        x, dy = rnd(rows, c, dtype=torch.float32, seed=120 + i), rnd(rows, c, dtype=torch.float32, seed=130 + i)
        g, b = rnd(c, dtype=torch.float32, seed=140 + i), rnd(c, dtype=torch.float32, seed=150 + i)
        _, _, mean, rstd = K.layernorm_fwd(x, g, b, 1e-6, want_bf16=False, want_f32=True)
        dg1, db1 = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
        dx1, _ = K.layernorm_bwd(dy, x, g, b, mean, rstd, dgamma=dg1, dbeta=db1)
        dg2, db2 = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
        dx2, _ = K.layernorm_bwd(dy, x, g, b, mean, rstd, dgamma=dg2, dbeta=db2, deferred=items)
        assert torch.equal(dx1, dx2)
        direct.append((dg1, db1, dg2, db2))
    assert len(items) == 3
    K.ln_param_reduce_batched(items)
    for dg1, db1, dg2, db2 in direct:
        close(dg2, dg1, 1e-5, 1e-4 * float(dg1.abs().max()), "deferred dgamma")
        close(db2, db1, 1e-5, 1e-4 * float(db1.abs().max()), "deferred dbeta")


@pytest.mark.parametrize(("hd", "heads", "tq", "tk", "generic"), [
    (64, 3, 685, 685, False), (32, 4, 300, 77, False), (64, 2, 64, 64, False), (32, 2, 129, 200, False), (64, 2, 200, 130, True),
    (8, 2, 129, 129, True), (8, 2, 385, 128, True), (16, 4, 50, 33, False), (32, 16, 2053, 684, False)])
def test_attention_forward_backward(hd: int, heads: int, tq: int, tk: int, generic: bool) -> None:
    b, c = 2, hd * heads
    self_attn = tq == tk
    if self_attn:  # consume a fused [q | k | v] projection buffer in place (strided views)
        qkv = rnd(b, tq, 3 * c, seed=30)
        q, k, v = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]
    else:
        q, kv = rnd(b, tq, c, seed=31), rnd(b, tk, 2 * c, seed=32)
        k, v = kv[..., :c], kv[..., c:]
    scale = hd**-0.5
    qr, kr, vr = (t.float().detach().clone().requires_grad_(True) for t in (q, k, v))
    ref = attn_ref(qr, kr, vr, heads)
    o, lse = K.attention_fwd(q, k, v, heads, scale, force_generic=generic)
    close(o, ref, 1e-2, 1e-2, "attention fwd")  # bf16 P and bf16 output
    # lse check (log2 domain)
    s = (qr.detach().reshape(b, tq, heads, hd).transpose(1, 2) @ kr.detach().reshape(b, tk, heads, hd).transpose(1, 2).transpose(-1, -2)) * scale
    close(lse, torch.logsumexp(s, dim=-1) / math.log(2.0), 1e-4, 1e-3, "lse")
    d_o = rnd(b, tq, c, seed=33)
    ref.backward(d_o.float())
    if self_attn:
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv[..., :c], dqkv[..., c:2 * c], dqkv[..., 2 * c:]
    else:
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        dk, dv = dkv[..., :c], dkv[..., c:]
    K.attention_bwd(q, k, v, o, d_o, lse, heads, scale, dq, dk, dv, force_generic=generic)
    # gradients pass through bf16 P/dS and bf16 outputs: 2% of the tensor scale
    for name, got, want in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        tol = 2e-2 * float(want.abs().max())
        close(got, want, 2e-2, tol, f"attention {name}")


@pytest.mark.parametrize("hd", [32, 64])
def test_attention_forward_lazy_rescale_branch(hd: int, monkeypatch: pytest.MonkeyPatch) -> None:
    """The forward kernel raises its running maximum lazily (only when a tile's maximum exceeds it by more than 2^6) and, at head_dim 32, takes the softmax
    denominator from an all-ones MFMA.  Bounded random inputs take the rescale branch in the first tile only, so this test FORCES it later: small scores
    everywhere, one key in the 5th tile whose score is ~+30 (log2 domain) for half of the queries and ~-30 for the others (the vote is per wave: lanes whose
    maximum did not move ride along), and a second, smaller spike in a later tile that stays under the threshold.  Full-tensor fp32 reference; the eager form
    (CINEMA_ATTN_FWD_V2=0) must agree with the lazy one to rounding, and so must their log-sum-exp rows."""
    b, heads, tq, tk = 2, 3, 300, 700
    c = heads * hd
    g = torch.Generator().manual_seed(77)
    q = (torch.randn(b, tq, c, generator=g) * 0.3)
    k = (torch.randn(b, tk, c, generator=g) * 0.3)
    v = torch.randn(b, tk, c, generator=g)
    sign = torch.where(torch.arange(tq) % 2 == 0, 1.0, -1.0)[None, :, None]
    q[:, :, :4] = 2.0 * sign          # first 4 channels of head 0: +-2
    k[:, 300, :4] = 16.0 * hd**0.5 / 4  # key 300 (tile 4): score +-(4 * 2 * 16 sqrt(hd) / 4) / sqrt(hd) = +-32 -> ~+-46 in log2 units
    k[:, 520, :4] = 1.5 * hd**0.5 / 4   # key 520 (tile 8): +-3 -> under the 2^6 threshold relative to the running maximum of the "-" queries
    q, k, v = (t.to(torch.bfloat16).to(DEV) for t in (q, k, v))
    ref = attn_ref(q.float(), k.float(), v.float(), heads)
    s = (q.float().reshape(b, tq, heads, hd).transpose(1, 2) @ k.float().reshape(b, tk, heads, hd).transpose(1, 2).transpose(-1, -2)) * hd**-0.5
    outs = {}
    for v2 in ("1", "0"):
        monkeypatch.setenv("CINEMA_ATTN_FWD_V2", v2)
        o, lse = K.attention_fwd(q, k, v, heads, hd**-0.5)
        close(o, ref, 1e-2, 1e-2, f"attention fwd spike V2={v2}")
        close(lse, torch.logsumexp(s, dim=-1) / math.log(2.0), 1e-4, 2e-3, f"lse spike V2={v2}")
        assert bool(torch.isfinite(o.float()).all())
        outs[v2] = (o.float(), lse)
    assert float((outs["1"][0] - outs["0"][0]).abs().max()) <= 2e-2 and float((outs["1"][1] - outs["0"][1]).abs().max()) <= 2e-3


@pytest.mark.parametrize(("hd", "heads", "tq", "tk", "force_g"), [
    (32, 16, 2053, 684, 0), (32, 4, 70, 33, 0), (32, 2, 130, 768, 0), (32, 3, 64, 97, 0),
    # head_dim 64 (attn_bwd_onepass_mfma): the encoder of config 2 per (batch, head) - one workgroup, three passes over the keys - and split over three workgroups;
    # 1728 keys (config 5's key count) over seven workgroups; 3073 tokens (config 4) over thirteen workgroups of one pass and over two workgroups of seven passes;
    # single-pass shapes without scratch
    (64, 12, 685, 685, 1), (64, 12, 685, 685, 0), (64, 4, 300, 1728, 0), (64, 2, 3073, 3073, 0), (64, 2, 3073, 3073, 2), (64, 3, 70, 33, 0), (64, 2, 130, 256, 0),
    (64, 2, 97, 1537, 3)])
def test_attention_backward_one_pass_equals_two_kernel_form(hd: int, heads: int, tq: int, tk: int, force_g: int, monkeypatch) -> None:  # noqa: ANN001
    """The one-pass attention backward kernels - attn_bwd_fused_mfma<32> (one workgroup per (batch, head), <= 768 keys, dQ through the in-LDS transpose of dS and a
    fixed-order cross-wave sum) and attn_bwd_onepass_mfma<64> (passes of 256 keys, dQ from shared dS tiles, running sums across passes, the keys of a (batch, head)
    split over G workgroups with a last-arriver sum in split order) - against fp32 autograd (2 % of the tensor scale: bf16 P / dS and outputs) and against the
    two-kernel form (same arithmetic up to the summation order of dQ and the bf16 rounding points: 1 % of the tensor scale), ragged key / query tails included;
    run twice (determinism); the arrival tickets are left at zero."""
    b = 2
    c = hd * heads
    q, kv = rnd(b, tq, c, seed=31), rnd(b, tk, 2 * c, seed=32)
    k, v = kv[..., :c], kv[..., c:]
    scale = hd**-0.5
    qr, kr, vr = (t.float().detach().clone().requires_grad_(True) for t in (q, k, v))
    ref = attn_ref(qr, kr, vr, heads)
    o, lse = K.attention_fwd(q, k, v, heads, scale)
    d_o = rnd(b, tq, c, seed=33)
    ref.backward(d_o.float())
    monkeypatch.setenv("CINEMA_ATTN_ONEPASS", "1")  # the head_dim 64 form is an option (measured slower than the kernel pair: off by default)
    if force_g:
        monkeypatch.setenv("CINEMA_ATTN_ONEPASS_G", str(force_g))
    outs = {}
    for form in ("1", "0", "1"):
        monkeypatch.setenv("CINEMA_ATTN_FUSED", form)
        dq, dkv = torch.full_like(q, float("nan")), torch.full_like(kv, float("nan"))
        K.attention_bwd(q, k, v, o, d_o, lse, heads, scale, dq, dkv[..., :c], dkv[..., c:])
        if form == "1" and "1" in outs:
            assert torch.equal(dq, outs["1"][0]) and torch.equal(dkv, outs["1"][1]), "one-pass backward is not deterministic"
        outs[form] = (dq, dkv)
    for name, i, sl, want in (("dq", 0, slice(None), qr.grad), ("dk", 1, slice(0, c), kr.grad), ("dv", 1, slice(c, 2 * c), vr.grad)):
        tol = float(want.abs().max())
        close(outs["1"][i][..., sl], want, 2e-2, 2e-2 * tol, f"one-pass {name} vs autograd")
        close(outs["1"][i][..., sl], outs["0"][i][..., sl], 0.0, 1e-2 * tol, f"one-pass {name} vs two-kernel form")
    assert all(int(t.abs().sum()) == 0 for t in K._ATTN_COUNTERS.values())  # noqa: SLF001


def test_attention_rescale_branch() -> None:
    """Force large running-max jumps between key tiles (guide rule 26): one spiked key per tile, increasing."""
    b, heads, hd, t = 1, 1, 64, 256
    q, k, v = rnd(b, t, hd, seed=34), rnd(b, t, hd, seed=35), rnd(b, t, hd, seed=36)
    k = k.clone()
    for tile in range(4):
        k[0, tile * 64 + 7] = q[0, 3] * (2.0 + 2.0 * tile)
    o, _ = K.attention_fwd(q, k, v, heads, hd**-0.5)
    close(o, attn_ref(q, k, v, heads), 1e-2, 2e-2, "attention with max jumps")


# ------------------------------------------------------------------------------------------------ depthwise conv
@pytest.mark.parametrize(("shape", "c"), [((2, 12, 10, 6), 64), ((1, 7, 9, 16), 128), ((2, 13, 11), 64), ((2, 6, 4, 5), 8), ((3, 6, 8), 8)])
def test_dwconv(shape: tuple, c: int) -> None:
    b, *sp = shape
    nd = len(sp)
    x = rnd(b, *sp, c, seed=40)
    w = rnd(c, 1, *([5] * nd), dtype=torch.float32, seed=41, scale=0.2)
    bias = rnd(c, dtype=torch.float32, seed=42)
    conv = F.conv3d if nd == 3 else F.conv2d
    xr = x.float().movedim(-1, 1).contiguous().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    ref = conv(xr, wr, br, padding=2, groups=c)
    y = K.dwconv_fwd(x, w, bias)
    close(y, ref.movedim(1, -1), 1e-2, 2e-2, "dwconv fwd")
    dy = rnd(b, *sp, c, seed=43)
    ref.backward(dy.float().movedim(-1, 1))
    close(K.dwconv_bwd_data(dy, w), xr.grad.movedim(1, -1), 1e-2, 2e-2, "dwconv dgrad")
    dw, db = torch.zeros_like(w), torch.zeros_like(bias)
    K.dwconv_bwd_weight(x, dy, dw, db)
    close(dw, wr.grad, 1e-3, 1e-3 * float(wr.grad.abs().max()) + 1e-3, "dwconv wgrad")
    close(db, br.grad, 1e-3, 1e-3 * float(br.grad.abs().max()) + 1e-3, "dwconv bgrad")


# ------------------------------------------------------------------------------------------------ patches / rows
@pytest.mark.parametrize("pair", [False, True])
@pytest.mark.parametrize(("tok_grid", "block", "c", "levels"), [((3, 3, 4), (4, 4, 1), 64, 2), ((3, 4), (2, 2), 128, 1), ((2, 3, 2), (2, 2, 2), 64, 1), ((3, 5), (4, 4), 64, 2),
                                                                ((4, 5, 7), (2, 2, 1), 128, 1), ((5, 4, 6), (4, 4, 1), 128, 2), ((6, 3, 2), (2, 2, 1), 64, 1), ((5, 6), (4, 4), 128, 2)])
def test_sparse_dwconv_matches_dense_masked_conv(tok_grid: tuple, block: tuple, c: int, levels: int, pair: bool, monkeypatch) -> None:  # noqa: ANN001
    """Visible-voxel depthwise conv (forward, data gradient, weight gradient) == the dense kernels applied to a volume
    whose masked voxels are zero, read back at the visible voxels (the identity the MAE stem relies on)."""
    from cinema_amd.convvit import hierarchical_positions

    # pair: the token-pair kernels of csrc/stem_dw.hip where they apply (every parametrisation except the 2x2x2 block); otherwise the neighbour-list kernels
    monkeypatch.setattr(K, "STEM_DW_PAIR", pair)
    nd = len(tok_grid)
    b = 2
    n_tok_all = math.prod(tok_grid)
    n_keep = max(2, n_tok_all // 4) if c == 64 else max(2, (2 * n_tok_all) // 5)
    g = torch.Generator(device="cpu").manual_seed(5)
    keep_pos = torch.stack([torch.randperm(n_tok_all, generator=g)[:n_keep].sort().values for _ in range(b)])
    keep = (torch.arange(b)[:, None] * n_tok_all + keep_pos).reshape(-1).to(torch.int32).to(DEV)
    rank = torch.full((b * n_tok_all,), -1, dtype=torch.int32, device=DEV)
    rank[keep.long()] = torch.arange(keep.numel(), dtype=torch.int32, device=DEV)
    if levels == 2:
        half = tuple(max(1, v // 2) for v in block)
        ps = [tuple(1 for _ in block), tuple(v // h for v, h in zip(block, half)), half]
    else:
        ps = [tuple(1 for _ in block), block]
    pos_l = hierarchical_positions(ps, 1)
    bv = math.prod(block)
    assert sorted(pos_l) == list(range(bv))
    pos = torch.tensor(pos_l, dtype=torch.int32, device=DEV)
    geom = K.sparse_geom(b, tok_grid, block, keep, rank, pos)
    # dense <-> compact index map
    spatial = tuple(t * k for t, k in zip(tok_grid, block))
    dense_of_row = torch.empty(keep.numel() * bv, dtype=torch.long)
    kp = keep.cpu().long()
    for r in range(kp.numel()):
        bb, t = int(kp[r]) // n_tok_all, int(kp[r]) % n_tok_all
        tc = []
        for gdim in reversed(tok_grid):
            tc.append(t % gdim)
            t //= gdim
        tc.reverse()
        for u in range(bv):
            uu, uc = u, []
            for bdim in reversed(block):
                uc.append(uu % bdim)
                uu //= bdim
            uc.reverse()
            vid = bb
            for d in range(nd):
                vid = vid * spatial[d] + tc[d] * block[d] + uc[d]
            dense_of_row[r * bv + pos_l[u]] = vid
    dense_of_row = dense_of_row.to(DEV)
    nvox = b * math.prod(spatial)
    ks = (5,) * nd
    w = rnd(c, 1, *ks, scale=0.2, dtype=torch.float32, seed=21)
    bias = rnd(c, dtype=torch.float32, seed=22)
    xc = rnd(keep.numel() * bv, c, seed=23)
    dyc = rnd(keep.numel() * bv, c, seed=24)
    xd = torch.zeros(nvox, c, dtype=torch.bfloat16, device=DEV)
    xd[dense_of_row] = xc
    dyd = torch.zeros(nvox, c, dtype=torch.bfloat16, device=DEV)
    dyd[dense_of_row] = dyc
    xd5, dyd5 = xd.view(b, *spatial, c), dyd.view(b, *spatial, c)
    # forward
    y_ref = K.dwconv_fwd(xd5, w, bias).view(nvox, c)[dense_of_row]
    y = K.sparse_dwconv(xc, w, bias, geom)
    close(y, y_ref, 1e-2, 1e-2, "sparse dwconv fwd")  # same fp32 taps, different summation order, one bf16 rounding
    # data gradient (taps flipped), read at visible voxels
    dx_ref = K.dwconv_bwd_data(dyd5, w).view(nvox, c)[dense_of_row]
    dx = K.sparse_dwconv(dyc, w, None, geom, flip=True)
    close(dx, dx_ref, 1e-2, 1e-2, "sparse dwconv bwd data")
    # weight gradient
    dw_ref, db_ref = torch.zeros_like(w), torch.zeros(c, device=DEV)
    K.dwconv_bwd_weight(xd5, dyd5, dw_ref, db_ref)
    dw, db = torch.zeros_like(w), torch.zeros(c, device=DEV)
    K.sparse_dwconv_bwd_weight(xc, dyc, tuple(w.shape), dw, db, geom)
    close(dw, dw_ref, 1e-3, 1e-2 * float(dw_ref.abs().max()), "sparse dwconv wgrad")
    close(db, db_ref, 1e-3, 1e-3 * float(db_ref.abs().max()), "sparse dwconv bias grad")
    # the default is the token-pipelined kernel on the halo index table; the per-token index chase gives the same sums (same order within a workgroup,
    # other workgroup chunks: fp32 rounding of the slab reduction only)
    kd = (1,) * (3 - nd) + ks
    if pair and K.load().cinema_stem_dw_supported(__import__("ctypes").byref(geom), c, *kd):
        assert not geom.halo_idx and not geom.nbr_lists  # the token-pair kernels ran: no list, no halo table was built
        dw3, db3 = torch.zeros_like(w), torch.zeros(c, device=DEV)
        K.sparse_dwconv_bwd_weight(xc, dyc, tuple(w.shape), dw3, db3, geom)
        assert torch.equal(dw, dw3) and torch.equal(db, db3)  # ordered slab reduce: bit-identical run to run
        return
    assert K.SPARSE_WGRAD_PIPE and (5,) * 3 in geom.halo_idx or (1, 5, 5) in geom.halo_idx
    prev, K.SPARSE_WGRAD_PIPE = K.SPARSE_WGRAD_PIPE, False
    try:
        dw2, db2 = torch.zeros_like(w), torch.zeros(c, device=DEV)
        K.sparse_dwconv_bwd_weight(xc, dyc, tuple(w.shape), dw2, db2, geom)
    finally:
        K.SPARSE_WGRAD_PIPE = prev
    close(dw, dw2, 1e-5, 1e-5 * float(dw_ref.abs().max()), "sparse dwconv wgrad: pipelined vs index chase")
    close(db, db2, 1e-5, 1e-5 * float(db_ref.abs().max()), "sparse dwconv bias grad: pipelined vs index chase")


def test_patch_gather_scatter_channels_first_image() -> None:
    img = rnd(2, 3, 8, 12, 4, dtype=torch.float32, seed=50)
    patch, grid = (4, 4, 1), (2, 3, 4)
    geom = K.patch_geom(2, 3, grid, patch, img.stride())
    out = K.patch_gather(img, geom, torch.float32)
    ref = O.patchify(img.cpu(), patch).reshape(-1, 48).to(DEV)
    assert torch.equal(out, ref)
    back = torch.zeros_like(img)
    K.patch_scatter(out, back, geom)
    assert torch.equal(back, img)
    img2 = rnd(2, 1, 8, 6, dtype=torch.float32, seed=51)
    geom2 = K.patch_geom(2, 1, (4, 3), (2, 2), img2.stride())
    assert torch.equal(K.patch_gather(img2, geom2, torch.float32), O.patchify(img2.cpu(), (2, 2)).reshape(-1, 4).to(DEV))


def test_patch_gather_channels_last_with_token_subset() -> None:
    b, c, sp = 2, 16, (8, 6, 4)
    x = rnd(b, *sp, c, dtype=torch.float32, seed=52)  # channels-last feature map
    patch, grid = (2, 2, 1), (4, 3, 4)
    strides = (x.stride(0), 1, x.stride(1), x.stride(2), x.stride(3))
    ref_all = O.patchify(x.movedim(-1, 1).cpu(), patch).reshape(-1, 4 * c).to(DEV)
    geom = K.patch_geom(b, c, grid, patch, strides)
    close(K.patch_gather(x, geom), ref_all, 1e-2, 1e-2, "gather all (bf16 out)")
    idx = torch.tensor([0, 5, 47, 48, 95, 17], dtype=torch.int32, device=DEV)
    gsub = K.patch_geom(b, c, grid, patch, strides, token_idx=idx)
    assert torch.equal(K.patch_gather(x, gsub, torch.float32), ref_all[idx.long()])
    dst = torch.zeros_like(x)
    K.patch_scatter(ref_all[idx.long()].contiguous(), dst, gsub)
    ref_dst = torch.zeros_like(ref_all)
    ref_dst[idx.long()] = ref_all[idx.long()]
    assert torch.equal(dst, O.unpatchify(ref_dst.reshape(b, -1, 4 * c).cpu(), patch, grid).movedim(1, -1).to(DEV))


@pytest.mark.parametrize(("shape", "pad_to", "permuted"), [((32, 5, 4, 4, 1), 1, True), ((16, 1, 3, 3, 3), 8, False), ((24, 8, 2, 2), 1, False)])
def test_patch_weight_relayout(shape: tuple, pad_to: int, permuted: bool) -> None:
    """Conv weight (out, c, *k) <-> GEMM operand rows in (*k, c) feature order: bf16 rounding only one way, exact fp32 adds the other way."""
    w = rnd(*shape, dtype=torch.float32, seed=70)
    kvol = math.prod(shape[2:])
    jmap = torch.randperm(kvol, generator=torch.Generator().manual_seed(3)).to(torch.int32).to(DEV) if permuted else None
    rows = K.patch_weight_rows(w, jmap=jmap, pad_to=pad_to)
    ref = w.reshape(shape[0], shape[1], kvol).permute(0, 2, 1)  # [out, kvol, c]
    if permuted:
        ref = ref[:, jmap.long(), :]
    ref = ref.reshape(shape[0], -1)
    f = ref.shape[1]
    assert rows.shape[1] == (f + pad_to - 1) // pad_to * pad_to
    assert torch.equal(rows[:, :f], ref.to(torch.bfloat16)) and not rows[:, f:].any()
    g_rows = rnd(shape[0], rows.shape[1], dtype=torch.float32, seed=71)
    acc = rnd(*shape, dtype=torch.float32, seed=72)
    expect = g_rows[:, :f].reshape(shape[0], kvol, shape[1])
    if permuted:
        inv = torch.empty_like(jmap)
        inv[jmap.long()] = torch.arange(kvol, dtype=torch.int32, device=DEV)
        expect = expect[:, inv.long(), :]
    expect = acc + expect.permute(0, 2, 1).reshape(shape)
    K.patch_weight_grad_accumulate(g_rows, acc, jmap)
    assert torch.equal(acc, expect)


@pytest.mark.parametrize(("b", "n", "ratio"), [(3, 144, 0.75), (2, 2304, 0.75), (4, 37, 0.5), (1, 5000, 0.9)])
def test_random_mask_and_selection_lists(b: int, n: int, ratio: float) -> None:
    """One-launch mask recipe == the reference recipe argsort(argsort(noise)) >= n_keep (cinema/mae/mae.py:30-65), incl. tied noise values;
    one-launch raster-ordered kept/dropped lists == boolean-mask indexing order (mae.py:550).  Integer work: bit-exact."""
    n_keep = int(n * (1 - ratio))
    g = torch.Generator(device="cpu").manual_seed(5)
    noise = torch.rand(b, n, generator=g)
    noise[:, n // 3] = noise[:, n // 2]  # a tie in every row: resolved by index, like a stable sort
    noise = noise.to(DEV)
    mask = K.random_mask(noise, n_keep)
    ref = torch.argsort(torch.argsort(noise, dim=1, stable=True), dim=1, stable=True) >= n_keep
    assert mask.dtype == torch.bool and torch.equal(mask, ref)
    keep_pos, drop_pos, keep, drop = K.mask_select(mask, n_keep)
    ar = torch.arange(n, device=DEV, dtype=torch.int32)[None].expand(b, -1)
    base = torch.arange(b, device=DEV, dtype=torch.int32)[:, None] * n
    assert torch.equal(keep_pos, ar[~ref]) and torch.equal(drop_pos, ar[ref])
    assert torch.equal(keep, (base + ar)[~ref]) and torch.equal(drop, (base + ar)[ref])


@pytest.mark.parametrize(("grid", "block"), [((3, 4, 2), (4, 4, 1)), ((5, 6), (2, 2))])
def test_visible_index(grid: tuple, block: tuple) -> None:
    """rank table and stage-1 voxel ids of the kept tokens vs the torch integer ops they replace (bit-exact)."""
    batch, n_tok_all, vol = 2, math.prod(grid), math.prod(block)
    g = torch.Generator(device="cpu").manual_seed(9)
    keep = torch.sort(torch.randperm(batch * n_tok_all, generator=g)[: batch * n_tok_all // 3]).values.to(torch.int32).to(DEV)
    inv1 = torch.randperm(vol, generator=g).to(torch.int32).to(DEV)
    rank, idx1 = K.visible_index(keep, batch, grid, block, inv1)
    ref_rank = torch.full((batch * n_tok_all,), -1, dtype=torch.int32, device=DEV)
    ref_rank[keep.long()] = torch.arange(keep.numel(), dtype=torch.int32, device=DEV)
    assert torch.equal(rank, ref_rank)
    grid1 = tuple(a * b_ for a, b_ in zip(grid, block))
    t, bb = keep.long() % n_tok_all, keep.long() // n_tok_all
    tc = []
    for gd in reversed(grid):
        tc.append(t % gd)
        t = t // gd
    tc.reverse()
    u, uc = inv1.long(), []
    for bd in reversed(block):
        uc.append(u % bd)
        u = u // bd
    uc.reverse()
    vid = bb[:, None]
    for d in range(len(grid)):
        vid = vid * grid1[d] + (tc[d][:, None] * block[d] + uc[d][None, :])
    assert torch.equal(idx1, vid.reshape(-1).to(torch.int32))


def test_row_copy_cast_transpose_gelu() -> None:
    src = rnd(10, 32, dtype=torch.float32, seed=60)
    add = rnd(7, 32, dtype=torch.float32, seed=61)
    dst = torch.zeros(12, 32, dtype=torch.float32, device=DEV)
    di = torch.tensor([11, 0, 3], dtype=torch.int32, device=DEV)
    si = torch.tensor([2, 9, 4], dtype=torch.int32, device=DEV)
    ai = torch.tensor([6, 6, 1], dtype=torch.int32, device=DEV)
    K.row_copy(dst, src, dst_idx=di, src_idx=si, add=add, add_idx=ai)
    ref = torch.zeros_like(dst)
    ref[di.long()] = src[si.long()] + add[ai.long()]
    assert torch.equal(dst, ref)
    K.row_copy(dst, src, dst_idx=di, src_idx=si, accumulate=True)
    ref[di.long()] += src[si.long()]
    assert torch.equal(dst, ref)
    d16 = torch.zeros(3, 32, dtype=torch.bfloat16, device=DEV)
    K.row_copy(d16, None, add=add, add_idx=ai)
    assert torch.equal(d16, add[ai.long()].to(torch.bfloat16))
    odd = rnd(5, 7, dtype=torch.float32, seed=62)
    assert torch.equal(K.row_copy(torch.zeros(5, 7, dtype=torch.float32, device=DEV), odd), odd)
    x = rnd(1003, dtype=torch.float32, seed=63)
    assert torch.equal(K.cast(x, torch.bfloat16), x.to(torch.bfloat16))
    assert torch.equal(K.cast(x.to(torch.bfloat16), torch.float32), x.to(torch.bfloat16).float())
    w = rnd(70, 130, dtype=torch.float32, seed=64)
    assert torch.equal(K.transpose_cast(w), w.t().contiguous().to(torch.bfloat16))
    h = rnd(999, seed=65, scale=2.0)
    close(K.gelu_fwd(h), F.gelu(h.float()), 1e-2, 1e-2, "gelu")
    hx = h.float().requires_grad_(True)
    dy = rnd(999, seed=66)
    F.gelu(hx).backward(dy.float())
    close(K.gelu_bwd(h, dy), hx.grad, 1e-2, 1e-2, "gelu bwd")


# ------------------------------------------------------------------------------------------------ loss / optimiser
@pytest.mark.parametrize("norm_target", [False, True])
def test_masked_mse(norm_target: bool) -> None:
    b = 2
    img = torch.rand(b, 1, 32, 32, 4, generator=torch.Generator().manual_seed(70)).to(DEV)
    patch, grid = (16, 16, 1), (2, 2, 4)
    mask = O.random_patch_mask(b, 16, 0.75, torch.Generator().manual_seed(71))
    ids = torch.nonzero(mask.flatten()).flatten().to(torch.int32).to(DEV)
    pred = rnd(ids.numel(), 256, seed=72)
    target = O.patchify(img.cpu(), patch)
    pr = pred.float().cpu().reshape(b, -1, 256).requires_grad_(True)
    ref_loss, ref_metrics = O.mse_loss(target, pr, mask, norm_target)
    geom = K.patch_geom(b, 1, grid, patch, img.stride(), token_idx=ids)
    loss = torch.zeros(1, dtype=torch.float32, device=DEV)
    K.mse_fwd(img, geom, pred, norm_target, 1e-6, loss)
    close(loss.cpu(), ref_loss.detach().reshape(1), 1e-4, 1e-6, "mse loss")
    stats = torch.zeros(2, dtype=torch.float32, device=DEV)
    K.patch_stats(img, K.patch_geom(b, 1, grid, patch, img.stride()), stats)
    close(stats.cpu(), torch.stack([ref_metrics["target_mean"], ref_metrics["target_std"]]), 1e-4, 1e-6, "patch stats")
    ref_loss.backward()
    up = torch.full((1,), 0.5, dtype=torch.float32, device=DEV)
    dpred = K.mse_bwd(img, geom, pred, norm_target, 1e-6, up, 1.0 / pred.numel())
    close(dpred.cpu(), 0.5 * pr.grad.reshape(-1, 256), 1e-2, 1e-7, "mse dpred")


def test_mean_finite() -> None:
    vals = torch.tensor([1.0, float("nan"), 3.0, float("inf")], device=DEV)
    mean, coef = torch.zeros(1, device=DEV), torch.zeros(4, device=DEV)
    K.mean_finite(vals, mean, coef)
    assert float(mean) == 2.0 and coef.tolist() == [0.5, 0.0, 0.5, 0.0]
    K.mean_finite(torch.tensor([float("nan")], device=DEV), mean, None)
    assert math.isnan(float(mean))


def test_adamw_and_clip() -> None:
    n = 10007
    p0, g = rnd(n, dtype=torch.float32, seed=80), rnd(n, dtype=torch.float32, seed=81)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    sq = torch.zeros(1, device=DEV)
    K.sqnorm(g, sq)
    close(sq, (g.double() ** 2).sum().float().reshape(1), 1e-5, 0, "sqnorm")
    coef, norm = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    K.clip_coef(sq, 5.0, coef, norm)
    p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    shadow = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    for step in (1, 2, 3):
        pr.grad = g.clone()
        ref_norm = torch.nn.utils.clip_grad_norm_([pr], 5.0)
        opt.step()
        K.adamw(p, g, m, v, 1e-3, 0.9, 0.95, 1e-8, 0.05, step, clip=coef, shadow=shadow)
    close(norm, ref_norm.reshape(1), 1e-5, 0, "grad norm")
    close(p, pr.detach(), 1e-5, 1e-6, "adamw params")
    assert torch.equal(shadow, p.to(torch.bfloat16))


@pytest.mark.parametrize(("hd", "heads", "tq", "tk"), [(64, 4, 200, 200), (32, 4, 300, 150), (64, 2, 685, 685)])
def test_attention_backward_with_the_second_half_of_the_output(hd: int, heads: int, tq: int, tk: int) -> None:
    """delta = rowsum(dO O) from BOTH bf16 halves of the forward output (``attention_fwd(want_lo=True)``): o + o_lo reproduces the fp32 output to 2^-16, and with keys
    that share a large common component and a flat softmax (the late blocks of a deep encoder: |mean key| = 5 x the spread) dQ is right to ~1 %, where delta from
    the bf16 output alone is off by several per cent (the error eps of delta shifts every dS of a query by P eps and comes back multiplied by sum_j P_ij K_j)."""
    b, c = 2, heads * hd
    g = torch.Generator().manual_seed(7)
    q = (torch.randn(b, tq, c, generator=g) * 0.3).to(torch.bfloat16).to(DEV)
    k = (torch.randn(b, tk, c, generator=g) * 0.3 + 1.5 * torch.randn(1, 1, c, generator=g)).to(torch.bfloat16).to(DEV)
    v = torch.randn(b, tk, c, generator=g).to(torch.bfloat16).to(DEV)
    d_o = torch.randn(b, tq, c, generator=g).to(torch.bfloat16).to(DEV)
    scale = hd ** -0.5
    o, lse, o_lo = K.attention_fwd(q, k, v, heads, scale, want_lo=True)

    def split(t: torch.Tensor) -> torch.Tensor:
        return t.double().reshape(b, -1, heads, hd).transpose(1, 2)

    qd, kd, vd, dod = split(q).requires_grad_(True), split(k), split(v), split(d_o)
    p = torch.softmax(qd @ kd.transpose(-1, -2) * scale, dim=-1)
    ref_o = p @ vd
    # o_lo is what the bf16 rounding of the kernel's fp32 output removed: at most half a bf16 ulp of o, and adding it back brings the output closer to the float64
    # value (the remaining difference is the bf16 rounding of P inside the P V product)
    assert bool((split(o_lo).abs() <= 2.0 ** -8 * split(o).abs() + 1e-30).all())
    err_hi, err_both = float((split(o) - ref_o).abs().mean()), float((split(o) + split(o_lo) - ref_o).abs().mean())
    assert err_both <= 0.7 * err_hi, (err_both, err_hi)
    (ref_o * dod).sum().backward()
    ref_dq = qd.grad

    def run(lo):  # noqa: ANN001, ANN202
        dq, dk, dv = (torch.empty_like(t) for t in (q, k, v))
        K.attention_bwd(q, k, v, o, d_o, lse, heads, scale, dq, dk, dv, o_lo=lo)
        return float((split(dq) - ref_dq).norm() / ref_dq.norm())

    with_lo, without = run(o_lo), run(None)
    print(f"dQ rel-L2 with both halves {with_lo:.3e}, with the bf16 output alone {without:.3e}")
    # (what remains is the bf16 rounding of P inside the FORWARD's P V product, which the recomputed P of the backward pass does not share: at a real late block of
    # ViT-Large the second half takes dQ from 13 % to 1.7 % and the q-weight gradient from 5.2 % to 0.2 %, tools/attn_dq_error.py)
    assert with_lo <= 2e-2 and with_lo <= 0.9 * without, (with_lo, without)


def test_segmentation_loss_vs_the_pinned_second_opinion_vectors() -> None:
    """CE + Dice of the HIP kernels (``cinema_seg_loss_fwd`` through ``_segmentation_loss``) against ``tests/golden/second_opinion.safetensors``: values on which the
    oracle and an independent float64 loop-style statement of monai's ``DiceLoss(include_background=False, softmax=True)`` agree (absent class, ignored voxels,
    all-background volume, 2-D included)."""
    from conftest import load_golden

    from cinema_amd.segmentation.train import _segmentation_loss

    g = load_golden("second_opinion.safetensors")
    for name in sorted({k.split("/")[1] for k in g if k.startswith("seg/")}):
        logits, labels, want = g[f"seg/{name}/logits"].to(DEV), g[f"seg/{name}/labels"].long().to(DEV), g[f"seg/{name}/values"]
        loss, m = _segmentation_loss(logits, labels)
        for i, k in enumerate(("cross_entropy", "mean_dice_loss", "loss")):
            assert abs(float(m[k]) - float(want[i])) <= 2e-5 * max(1.0, abs(float(want[i]))), (name, k, float(m[k]), float(want[i]))


@pytest.mark.parametrize(("shape", "c"), [((2, 24, 20, 6), 4), ((3, 40, 33), 3), ((1, 7, 5, 3), 2)])
def test_segmentation_loss_vs_oracle(shape: tuple, c: int) -> None:
    """CE(ignore -1) + soft Dice (reference cinema/segmentation/train.py:77-103) and its gradient against the fp32 oracle (autograd)."""
    from cinema_amd.segmentation.train import _segmentation_loss

    b, *sp = shape
    g = torch.Generator(device="cpu").manual_seed(17)
    logits = (torch.randn(b, c, *sp, generator=g) * 2.0)
    labels = torch.randint(-1, c, (b, 1, *sp), generator=g)
    ref_in = logits.clone().requires_grad_(True)
    ref_loss, ref_m = O.segmentation_loss_one_view(ref_in, labels)
    (ref_loss * 1.7).backward()
    x = logits.to(DEV).requires_grad_(True)
    loss, m = _segmentation_loss(x, labels.to(DEV))
    (loss * 1.7).backward()
    for k in ("cross_entropy", "mean_dice_loss", "loss"):
        assert float(m[k]) == pytest.approx(float(ref_m[k]), rel=2e-5, abs=1e-6), k  # fp32 both sides, different summation order
    close(x.grad, ref_in.grad.to(DEV), 1e-3, 1e-7, "seg loss grad")


# ------------------------------------------------------------------------------------------------ fp8 forward GEMM (BASELINE config 5)
def _e4m3_decode(u: torch.Tensor) -> torch.Tensor:
    """OCP e4m3fn bytes -> fp32 (bias 7, no infinities, 0x7f / 0xff = NaN), independent of any torch fp8 dtype support."""
    u = u.to(torch.int32)
    sign = torch.where((u & 0x80) != 0, -1.0, 1.0)
    e, m = (u >> 3) & 0xF, (u & 7).float()
    val = torch.where(e == 0, m / 8.0 * 2.0 ** -6, (1.0 + m / 8.0) * torch.pow(2.0, (e - 7).float()))
    return sign * val


def test_quantize_fp8_and_fp8_gemm_vs_dequantised_reference() -> None:
    """cinema_quantize_fp8: scale = amax / 448, every value within half an e4m3 step of x / scale; cinema_gemm_fp8 (MX-scaled MFMA with unit block
    scales) == fp32 matmul of the DEQUANTISED operands (the products of e4m3 values are exact in fp32, only the accumulation order differs), with
    the bias / GELU + pre-activation / fp32-residual epilogues; a transpose-detecting (asymmetric) operand pair."""
    m, n, k = 300, 256, 384
    a, w = rnd(m, k, seed=50, scale=2.0), rnd(n, k, seed=51, scale=0.5)
    a8, sa = K.quantize_fp8(a)
    w8, sw = K.quantize_fp8(w)
    assert float(sa) == pytest.approx(float(a.float().abs().max()) / 448.0, rel=1e-6)
    da, dw = _e4m3_decode(a8.cpu()) * float(sa), _e4m3_decode(w8.cpu()) * float(sw)
    assert float((da - a.float().cpu()).abs().max()) <= float(a.float().abs().max()) / 448.0 * 16.0 + 1e-6  # top binade: step 32 -> half step 16
    rel = (da - a.float().cpu()).abs() / a.float().cpu().abs().clamp_min(float(sa) * 2.0 ** -6)
    assert float(rel.max()) <= 0.0626  # 3 mantissa bits: relative rounding error <= 2^-4
    ref = da @ dw.t()
    got = K.gemm_fp8(a8, sa, w8, sw, out_dtype=torch.float32)
    close(got, ref.to(DEV), 1e-3, 2e-5 * float(ref.abs().max()), "fp8 gemm fp32 out")
    bias = rnd(n, dtype=torch.float32, seed=52)
    h = torch.empty(m, n, dtype=torch.bfloat16, device=DEV)
    y = K.gemm_fp8(a8, sa, w8, sw, bias=bias, act=1, aux_out=h)
    pre = ref.to(DEV) + bias
    close(h, pre, 1e-2, 1e-2 * float(pre.abs().max()), "fp8 gemm pre-activation")
    close(y, torch.nn.functional.gelu(pre), 1e-2, 1e-2 * float(pre.abs().max()), "fp8 gemm gelu")
    res = rnd(m, n, dtype=torch.float32, seed=53)
    z = K.gemm_fp8(a8, sa, w8, sw, bias=bias, residual=res)
    close(z, pre + res, 1e-3, 1e-4 * float(pre.abs().max()), "fp8 gemm residual")
    # the real config-5 shape of one encoder projection (ViT-Large: 13832 x 1024 x 1024): against the bf16 kernel on the same operands
    big_a, big_w = rnd(13832, 1024, seed=54), rnd(1024, 1024, seed=55, scale=0.03)
    qa, qsa = K.quantize_fp8(big_a)
    qw, qsw = K.quantize_fp8(big_w)
    y8, y16 = K.gemm_fp8(qa, qsa, qw, qsw, out_dtype=torch.float32), K.gemm(big_a, big_w, out_dtype=torch.float32)
    assert float((y8 - y16).norm() / y16.norm()) <= 6e-2  # two e4m3-quantised operands: ~2 x 2^-4 / sqrt(3) relative per product, averaged over K


def test_fp8_per_row_scales_and_layernorm_fused_quantisation() -> None:
    """Per-row (per-token) activation scaling: cinema_quantize_fp8_rows, the GEMM epilogue's row-scale vector, and the e4m3 copy written by the
    LayerNorm forward itself (same bytes / scales as quantising its bf16 output per row would give, up to the bf16 rounding the copy skips)."""
    m, n, k = 333, 128, 256
    a = rnd(m, k, seed=60) * (torch.arange(m, device=DEV).float()[:, None] * 0.05 + 0.1).bfloat16()  # rows of very different magnitude
    w = rnd(n, k, seed=61, scale=0.3)
    a8, sa = K.quantize_fp8_rows(a.contiguous())
    w8, sw = K.quantize_fp8(w)
    assert sa.shape == (m,) and torch.allclose(sa.cpu(), a.float().abs().amax(1).cpu() / 448.0, rtol=1e-6)
    da = _e4m3_decode(a8.cpu()) * sa.cpu()[:, None]
    ref = da @ (_e4m3_decode(w8.cpu()) * float(sw)).t()
    got = K.gemm_fp8(a8, sa, w8, sw, out_dtype=torch.float32)
    close(got, ref.to(DEV), 1e-3, 2e-5 * float(ref.abs().max()), "fp8 gemm with row scales")
    rel_rows = ((da - a.float().cpu()).norm(dim=1) / a.float().cpu().norm(dim=1))
    assert float(rel_rows.max()) <= 0.04  # every row keeps e4m3's relative precision whatever its magnitude
    x = rnd(m, k, dtype=torch.float32, seed=62, scale=3.0) + 1.0
    gamma, beta = rnd(k, dtype=torch.float32, seed=63) * 0.2 + 1.0, rnd(k, dtype=torch.float32, seed=64) * 0.1
    y16, _, mean, rstd, (y8, rs) = K.layernorm_fwd(x, gamma, beta, 1e-5, want_fp8=True)
    y16b, _, mean_b, rstd_b = K.layernorm_fwd(x, gamma, beta, 1e-5)
    assert torch.equal(y16, y16b) and torch.equal(mean, mean_b) and torch.equal(rstd, rstd_b)
    yf = torch.nn.functional.layer_norm(x, (k,), gamma, beta, 1e-5)
    assert torch.allclose(rs, yf.abs().amax(1) / 448.0, rtol=1e-4)
    dq = (_e4m3_decode(y8.cpu()) * rs.cpu()[:, None])
    assert float(((dq - yf.cpu()).norm(dim=1) / yf.cpu().norm(dim=1)).max()) <= 0.04


def test_fp8_data_gradient_operands_transposed_shadows_and_gelu_grad_epilogue() -> None:
    """The fp8 data-gradient GEMM dX = dY W: cinema_quantize_fp8_segments_t writes, for every 2-D segment, exactly the transpose of the bytes
    cinema_quantize_fp8_segments wrote (same scale); cinema_gemm_fp8 on (per-row quantised dY, transposed shadow) == fp32 matmul of the dequantised
    operands, plain and through the x GELU'(pre-activation) epilogue of fc2's data gradient."""
    shapes = [(256, 128), (384, 264), (64, 512)]  # [out][in] weights; 264: a 64-wide tile edge in both directions
    flat = torch.cat([rnd(n, k, seed=120 + i, scale=0.3 + 0.2 * i).reshape(-1) for i, (n, k) in enumerate(shapes)])
    offs = [0]
    for n, k in shapes:
        offs.append(offs[-1] + n * k)
    bounds = torch.tensor([[offs[i], offs[i + 1]] for i in range(len(shapes))], dtype=torch.int64, device=DEV)
    desc = torch.tensor([[offs[i], n, k] for i, (n, k) in enumerate(shapes)], dtype=torch.int64, device=DEV)
    y, yt = torch.zeros(flat.numel(), dtype=torch.uint8, device=DEV), torch.zeros(flat.numel(), dtype=torch.uint8, device=DEV)
    scales = torch.ones(len(shapes), dtype=torch.float32, device=DEV)
    K.quantize_fp8_segments(flat, bounds, y, scales)
    K.quantize_fp8_segments_t(flat, desc, scales, yt)
    for i, (n, k) in enumerate(shapes):
        assert torch.equal(yt[offs[i]:offs[i + 1]].view(k, n), y[offs[i]:offs[i + 1]].view(n, k).t()), f"segment {i}"
    n, k = shapes[1]
    m = 333
    dy = rnd(m, n, seed=130) * (torch.arange(m, device=DEV).float()[:, None] * 0.01 + 0.05).bfloat16()
    d8, srow = K.quantize_fp8_rows(dy.contiguous())
    wt8 = yt[offs[1]:offs[2]].view(k, n)
    ddy = _e4m3_decode(d8.cpu()) * srow.cpu()[:, None]
    dwt = _e4m3_decode(wt8.cpu()) * float(scales[1])
    ref = (ddy @ dwt.t()).to(DEV)  # [m, k] = dY W
    close(K.gemm_fp8(d8, srow, wt8, scales[1:2], out_dtype=torch.float32), ref, 1e-3, 2e-5 * float(ref.abs().max()), "fp8 dgrad")
    pre = rnd(m, k, seed=131)
    gp = 0.5 * (1 + torch.erf(pre.float() / math.sqrt(2))) + pre.float() * torch.exp(-0.5 * pre.float() ** 2) / math.sqrt(2 * math.pi)
    close(K.gemm_fp8(d8, srow, wt8, scales[1:2], gelu_in=pre), ref * gp, 1e-2, 1e-2 * float(ref.abs().max()), "fp8 dgrad x gelu'")
    exact = dy.float() @ flat[offs[1]:offs[2]].view(n, k).float()
    assert float((ref - exact).norm() / exact.norm()) <= 6e-2  # what the two e4m3 roundings cost against the bf16 operands


# ------------------------------------------------------------------------------------------------ implicit-GEMM convolution (ConvResBlock convs, config 4)
@pytest.mark.parametrize(("spatial", "c_in", "c_out", "ks"), [((10, 9, 5), 32, 64, (3, 3, 3)), ((12, 11), 64, 32, (3, 3)), ((6, 7, 4), 8, 16, (3, 3, 3)),
                                                              ((9, 8, 3), 128, 40, (3, 3, 1))])
def test_implicit_gemm_conv_forward_and_data_gradient(spatial: tuple, c_in: int, c_out: int, ks: tuple) -> None:
    """cinema_conv_gemm_bf16 (the A tiles gathered from the channels-last volume, no im2col matrix) against torch's conv on the same bf16-rounded
    operands: forward with bias (+ fp32 residual) and the data gradient (transposed tap offsets + cinema_conv_weight_dgrad weights); also == the
    im2col + GEMM path it replaces."""
    import torch.nn.functional as F  # noqa: N812

    b, nd = 2, len(spatial)
    x = rnd(b, *spatial, c_in, seed=70)
    w = rnd(c_out, c_in, *ks, dtype=torch.float32, seed=71, scale=0.2)
    bias = rnd(c_out, dtype=torch.float32, seed=72)
    conv = F.conv3d if nd == 3 else F.conv2d
    xc = x.float().movedim(-1, 1)
    wr = w.bfloat16().float()
    ref = conv(xc, wr, bias, padding=tuple(k // 2 for k in ks)).movedim(1, -1).reshape(-1, c_out)
    w16 = K.patch_weight_rows(w, pad_to=8)
    taps = K.conv_tap_table(c_in, ks, spatial, w16.shape[1], False, x.device)
    got = K.conv_gemm(x, w16, taps, out_dtype=torch.float32, bias=bias)
    close(got, ref, 2e-3, 2e-3 * float(ref.abs().max()), "implicit conv forward")
    cols = K.im2col(x, ks)
    close(got, K.gemm(cols, w16, bias=bias, out_dtype=torch.float32), 1e-4, 1e-4 * float(ref.abs().max()), "implicit conv == im2col + gemm")
    res = rnd(ref.shape[0], c_out, dtype=torch.float32, seed=73)
    close(K.conv_gemm(x, w16, taps, bias=bias, residual=res), ref + res, 2e-3, 2e-3 * float(ref.abs().max()), "implicit conv + residual")
    ybf = K.conv_gemm(x, w16, taps, bias=bias)
    assert ybf.dtype == torch.bfloat16
    close(ybf, ref, 1e-2, 1e-2 * float(ref.abs().max()), "implicit conv bf16 out")
    # data gradient: dx = conv_transpose(dy, w) = the same kernel on dy with the transposed weights and negated offsets
    dy = rnd(b, *spatial, c_out, seed=74)
    xg = xc.clone().requires_grad_(True)
    conv(xg, wr, None, padding=tuple(k // 2 for k in ks)).backward(dy.float().movedim(-1, 1))
    want = xg.grad.movedim(1, -1).reshape(-1, c_in)
    wt = K.conv_weight_dgrad(w)
    taps_t = K.conv_tap_table(c_out, ks, spatial, wt.shape[1], True, x.device)
    dx = K.conv_gemm(dy, wt, taps_t, out_dtype=torch.float32)
    close(dx, want, 2e-3, 2e-3 * float(want.abs().max()), "implicit conv data gradient")
    # weight gradient: dW[co][(tap, ci)] += dy^T im2col(x) with the column matrix gathered on the fly, bias gradient as fused row sums
    wg = xc.clone()
    wpar = wr.clone().requires_grad_(True)
    bpar = bias.clone().requires_grad_(True)
    conv(wg, wpar, bpar, padding=tuple(k // 2 for k in ks)).backward(dy.float().movedim(-1, 1))
    want_w = wpar.grad.reshape(c_out, c_in, -1).permute(0, 2, 1).reshape(c_out, -1)  # features (tap, ci)
    coords = K.conv_coord_table(b, spatial, x.device)
    dw = torch.full((c_out, w16.shape[1]), 0.5, dtype=torch.float32, device=DEV)
    db = torch.zeros(c_out, dtype=torch.float32, device=DEV)
    K.conv_wgrad(dy.reshape(-1, c_out), x, taps, coords, dw, split_k=3, a_rowsum=db)
    close(dw[:, :want_w.shape[1]] - 0.5, want_w, 3e-3, 3e-3 * float(want_w.abs().max()), "implicit conv weight gradient")
    assert float((dw[:, want_w.shape[1]:] - 0.5).abs().max()) == 0.0 if dw.shape[1] > want_w.shape[1] else True
    close(db, bpar.grad, 1e-3, 1e-3 * float(bpar.grad.abs().max()), "implicit conv bias gradient")


@pytest.mark.parametrize(("spatial", "c_in", "c_out", "zb"), [((10, 9, 8), 32, 32, 4), ((7, 6, 12), 16, 64, 2), ((5, 8, 4), 64, 24, 4), ((6, 5, 6), 8, 32, 2)])
def test_z_blocked_implicit_conv_matches_plain(spatial: tuple, c_in: int, c_out: int, zb: int) -> None:
    """The z-blocked form of the implicit 3x3x3 convolution (a GEMM row = zb consecutive z voxels, block-banded weights from cinema_conv_weight_zblock):
    forward (+ bias, + residual), data gradient and weight / bias gradient against torch's conv on the same bf16-rounded operands, and equal to the
    one-row-per-voxel form up to fp32 summation order."""
    import torch.nn.functional as F  # noqa: N812

    b, ks = 2, (3, 3, 3)
    x = rnd(b, *spatial, c_in, seed=90)
    w = rnd(c_out, c_in, *ks, dtype=torch.float32, seed=91, scale=0.2)
    bias = rnd(c_out, dtype=torch.float32, seed=92)
    xc, wr = x.float().movedim(-1, 1), w.bfloat16().float()
    ref = F.conv3d(xc, wr, bias, padding=1).movedim(1, -1).reshape(-1, c_out)
    w16 = K.patch_weight_rows(w, pad_to=8)
    wz, bz = K.conv_weight_zblock(w16, c_in, zb, False, bias)
    assert wz.shape == (zb * c_out, 9 * (zb + 2) * c_in) and torch.equal(bz, bias.repeat(zb))
    taps_z = K.conv_tap_table(c_in, ks, spatial, wz.shape[1], False, x.device, zb=zb)
    got = K.conv_gemm(x, wz, taps_z, out_dtype=torch.float32, bias=bz, zb=zb).view(-1, c_out)
    close(got, ref, 2e-3, 2e-3 * float(ref.abs().max()), "z-blocked conv forward")
    plain = K.conv_gemm(x, w16, K.conv_tap_table(c_in, ks, spatial, w16.shape[1], False, x.device), out_dtype=torch.float32, bias=bias)
    close(got, plain, 1e-4, 1e-4 * float(ref.abs().max()), "z-blocked == plain implicit conv")
    res = rnd(ref.shape[0], c_out, dtype=torch.float32, seed=93)
    close(K.conv_gemm(x, wz, taps_z, bias=bz, residual=res, zb=zb).view(-1, c_out), ref + res, 2e-3, 2e-3 * float(ref.abs().max()), "z-blocked conv + residual")
    # data gradient (z-blocking chosen by the INPUT channel count there; the test uses the same factor)
    dy = rnd(b, *spatial, c_out, seed=94)
    xg = xc.clone().requires_grad_(True)
    F.conv3d(xg, wr, None, padding=1).backward(dy.float().movedim(-1, 1))
    want = xg.grad.movedim(1, -1).reshape(-1, c_in)
    wt = K.conv_weight_dgrad(w)
    wtz, _ = K.conv_weight_zblock(wt, c_out, zb, True)
    taps_t = K.conv_tap_table(c_out, ks, spatial, wtz.shape[1], True, x.device, zb=zb)
    dx = K.conv_gemm(dy, wtz, taps_t, out_dtype=torch.float32, zb=zb).view(-1, c_in)
    close(dx, want, 2e-3, 2e-3 * float(want.abs().max()), "z-blocked conv data gradient")
    # weight / bias gradient: R = dy_groups^T im2col_zb(x), folded into dW (accumulating) and db
    wpar, bpar = wr.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    F.conv3d(xc, wpar, bpar, padding=1).backward(dy.float().movedim(-1, 1))
    want_w = wpar.grad.reshape(c_out, c_in, -1).permute(0, 2, 1).reshape(c_out, -1)
    coords = K.conv_coord_table(b, spatial, x.device, zb=zb)
    r = torch.empty(zb * c_out, wz.shape[1], dtype=torch.float32, device=DEV)
    rs = torch.zeros(zb * c_out, dtype=torch.float32, device=DEV)
    K.conv_wgrad(dy.reshape(-1, zb * c_out), x, taps_z, coords, r, split_k=3, a_rowsum=rs, zb=zb, accumulate=False)
    dw = torch.full((c_out, w16.shape[1]), 0.5, dtype=torch.float32, device=DEV)
    db = torch.full((c_out,), 0.25, dtype=torch.float32, device=DEV)
    K.conv_wgrad_zfold(r, c_out, c_in, zb, dw, rs, db)
    close(dw[:, :want_w.shape[1]] - 0.5, want_w, 3e-3, 3e-3 * float(want_w.abs().max()), "z-blocked conv weight gradient")
    close(db - 0.25, bpar.grad, 1e-3, 1e-3 * float(bpar.grad.abs().max()), "z-blocked conv bias gradient")


@pytest.mark.parametrize(("rows", "n", "k"), [(100000, 4, 32), (777, 3, 64), (4097, 8, 8)])
def test_thin_linear_kernels(rows: int, n: int, k: int) -> None:
    """The 4-class segmentation head as streaming kernels (cinema_thin_linear_fwd / bwd) against fp32 torch on the same bf16-rounded input."""
    x = rnd(rows, k, seed=80)
    w, b = rnd(n, k, dtype=torch.float32, seed=81, scale=0.3), rnd(n, dtype=torch.float32, seed=82)
    y = K.thin_linear_fwd(x, w, b)
    ref = x.float() @ w.t() + b
    close(y, ref, 1e-5, 1e-5 * float(ref.abs().max()), "thin linear forward")
    dy = rnd(rows, n, dtype=torch.float32, seed=83)
    dw, db = torch.full((n, k), 1.0, device=DEV), torch.zeros(n, device=DEV)
    dx = K.thin_linear_bwd(x, w, dy, dw, db, want_dx=True)
    close(dx, dy @ w, 1e-2, 1e-2 * float((dy @ w).abs().max()), "thin linear dx")
    want_w = dy.t() @ x.float()
    close(dw - 1.0, want_w, 1e-3, 2e-4 * float(want_w.abs().max()) * (rows ** 0.5) / 30 + 1e-3, "thin linear dw")
    close(db, dy.sum(0), 1e-3, 1e-4 * rows ** 0.5, "thin linear db")
    assert K.thin_linear_bwd(x, w, dy, None, None, want_dx=False) is None


@pytest.mark.parametrize(("rows", "n", "k"), [(100003, 32, 1), (777, 64, 8), (4097, 4, 3), (50, 16, 2)])
def test_fanout_linear_kernels(rows: int, n: int, k: int) -> None:
    """Layers with at most 8 inputs as streaming kernels (cinema_fanout_linear_fwd / bwd; the 1 -> 32 channel shortcut of the raw-image ConvResBlock) against
    fp32 torch on the same bf16-rounded input."""
    x = rnd(rows, k, seed=95)
    w, b = rnd(n, k, dtype=torch.float32, seed=96, scale=0.3), rnd(n, dtype=torch.float32, seed=97)
    y = K.fanout_linear_fwd(x, w, b)
    ref = x.float() @ w.t() + b
    close(y, ref, 1e-5, 1e-5 * float(ref.abs().max()), "fanout fwd")
    dy = rnd(rows, n, dtype=torch.float32, seed=98)
    dw = torch.full((n, k), 0.5, dtype=torch.float32, device=DEV)
    db = torch.full((n,), 0.25, dtype=torch.float32, device=DEV)
    dx = K.fanout_linear_bwd(x, w, dy, dw, db, want_dx=True)
    want_dw, want_db, want_dx = dy.t() @ x.float(), dy.sum(0), dy @ w
    close(dw - 0.5, want_dw, 2e-4, 2e-4 * float(want_dw.abs().max()) + 1e-3, "fanout dW")
    close(db - 0.25, want_db, 2e-4, 2e-4 * float(want_db.abs().max()) + 1e-3, "fanout db")
    close(dx, want_dx, 1e-2, 1e-2 * float(want_dx.abs().max()), "fanout dx (bf16)")
    assert K.fanout_linear_bwd(x, w, dy, dw, None, want_dx=False) is None


@pytest.mark.parametrize(("spatial", "ks", "n"), [((9, 10, 6), (3, 3, 3), 32), ((12, 11), (3, 3), 16), ((5, 6, 4), (3, 3, 1), 8), ((7, 5, 3), (1, 1, 1), 64)])
def test_one_channel_stencil_conv(spatial: tuple, ks: tuple, n: int) -> None:
    """cinema_conv1ch_fwd / bwd (the first conv of the raw-image ConvResBlock) against torch's conv autograd on the same bf16-rounded input, fp32 weights."""
    import torch.nn.functional as F  # noqa: N812

    b, nd = 2, len(spatial)
    x = rnd(b, *spatial, seed=110)
    w = rnd(n, 1, *ks, dtype=torch.float32, seed=111, scale=0.3)
    bias = rnd(n, dtype=torch.float32, seed=112)
    conv = F.conv3d if nd == 3 else F.conv2d
    xr = x.float().unsqueeze(1).requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    ref = conv(xr, wr, br, padding=tuple(k // 2 for k in ks))
    y = K.conv1ch_fwd(x, w, bias)
    want = ref.movedim(1, -1).reshape(-1, n)
    close(y, want, 1e-5, 1e-5 * float(want.abs().max()), "conv1ch fwd")
    dy = rnd(y.shape[0], n, dtype=torch.float32, seed=113)
    ref.backward(dy.reshape(b, *spatial, n).movedim(-1, 1))
    dw = torch.full_like(w, 0.5)
    db = torch.full_like(bias, 0.25)
    dx = K.conv1ch_bwd(x, w, dy, dw, db, want_dx=True)
    close(dw - 0.5, wr.grad, 3e-4, 3e-4 * float(wr.grad.abs().max()) + 1e-3, "conv1ch dW")
    close(db - 0.25, br.grad, 3e-4, 3e-4 * float(br.grad.abs().max()) + 1e-3, "conv1ch db")
    close(dx.reshape(b, *spatial), xr.grad[:, 0], 1e-2, 1e-2 * float(xr.grad.abs().max()), "conv1ch dx (bf16)")
    assert K.conv1ch_bwd(x, w, dy, None, None, want_dx=False) is None


@pytest.mark.parametrize("rows", [704, 5000, 13824])
def test_gemm_fp8_wgrad_p256_vs_independent_e4m3_decoder(rows: int) -> None:
    """cinema_gemm_fp8_wgrad_p256: weight gradients on ROW-MAJOR [token][feature] e4m3 operands (per-tensor scales), fragments by ds_read_b64_tr_b8, the MX-scaled
    MFMA with unit block scales, k-slices finished inside the persistent launch.  Reference: fp32 matmul of the operands decoded by an independent e4m3
    decoder (products of e4m3 values are exact in fp32: only the accumulation order differs -> 1e-3 of the largest element); ragged feature counts (not multiples
    of 256), a ragged last 64-token phase (rows % 64 != 0), accumulation into an existing gradient, twice on the same workspace (counters back to zero).
    Also against the bf16 weight gradient of the unquantised operands: relative L2 <= 6e-2 (two e4m3 operands, see test_fp8_quantise_and_gemm)."""
    shapes = [(512, 256), (256, 768), (80, 208), (272, 512)] if rows < 10000 else [(4096, 1024), (1024, 4096), (1024, 1024)]  # a ViT-Large block's fc1, fc2, proj
    for rep in range(2):
        probs, ref, ref16 = [], [], []
        for i, (n, k) in enumerate(shapes):
            dy = rnd(rows, n, scale=0.5, seed=180 + i + 7 * rep)
            x = rnd(rows, k, scale=0.5, seed=190 + i + 7 * rep)
            dy8, sdy = K.quantize_fp8(dy)
            x8, sx = K.quantize_fp8(x)
            base = rnd(n, k, dtype=torch.float32, seed=200 + i)
            probs.append((dy8, sdy, x8, sx, base.clone()))
            ddy, dx = _e4m3_decode(dy8.cpu()).to(DEV) * sdy, _e4m3_decode(x8.cpu()).to(DEV) * sx
            ref.append(base + ddy.t() @ dx)
            ref16.append(dy.float().t() @ x.float())
        K.gemm_fp8_wgrad_grouped(probs)
        for (dy8, sdy, x8, sx, dst), rd, r16, (n, k) in zip(probs, ref, ref16, shapes):
            close(dst, rd, 0.0, 1e-3 * float(rd.abs().max()), f"fp8 wgrad {n}x{k} rep {rep}")
            base = rnd(n, k, dtype=torch.float32, seed=200 + shapes.index((n, k)))
            assert float(((dst - base) - r16).norm() / r16.norm()) <= 6e-2
    ws = K._p256_workspace(torch.device(DEV, torch.cuda.current_device()))
    assert int(ws[:16384].view(torch.int32).abs().sum()) == 0, "tile counters / error word not left at zero"
    with pytest.raises(K.HipLibraryError):  # feature counts must be multiples of 16 bytes
        K.gemm_fp8_wgrad_grouped([(torch.zeros(64, 24, dtype=torch.uint8, device=DEV), torch.ones(1, device=DEV), torch.zeros(64, 32, dtype=torch.uint8, device=DEV),
                                   torch.ones(1, device=DEV), torch.zeros(24, 32, device=DEV))])


def test_q8_delayed_scaling_producers() -> None:
    """8-bit output copies with per-tensor DELAYED scaling (``cinema_q8_out``): a site's first launch only records the maximum; ``cinema_fp8_sites_update`` turns it
    into scale = margin * amax / 448; from then on the producers - stand-alone pass, LayerNorm forward / backward, the GEMM epilogues (plain bf16, GELU,
    x GELU') of the bf16 and the e4m3 kernels - write e4m3(sat(value / scale)): decoded with an independent e4m3 decoder every copy is within half an e4m3
    step (2^-4 relative, 2^-10 of the scale absolute) of the bf16 output of the same launch; values beyond the previous maximum saturate instead of overflowing."""
    from cinema_amd import tape as T

    sites = T.Fp8Sites(torch.device(DEV, torch.cuda.current_device()))

    def check(q8, ref16, what):  # noqa: ANN001, ANN202
        y8, sc = q8
        dec = _e4m3_decode(y8.cpu()).to(DEV) * sc
        err = (dec - ref16.float()).abs()
        tol = ref16.float().abs() * 2.0 ** -4 + float(sc) * 2.0 ** -9
        assert bool((err <= tol).all()), (what, float((err - tol).max()))

    x = rnd(3000, 512, scale=2.0, seed=301)
    s0 = sites.site(("alone", 0))
    assert not s0.ready and K.quantize_fp8_site(x, s0) is None
    sites.update()
    assert s0.ready and abs(float(s0.scale) - T.FP8_MARGIN * float(x.float().abs().max()) / 448.0) <= 1e-6 * float(s0.scale)
    check(K.quantize_fp8_site(x, s0), x, "stand-alone")
    big = K.quantize_fp8_site((x.float() * 4).to(torch.bfloat16), s0)  # 4 x the recorded maximum: saturates at 448 x scale, no NaN byte
    assert int(((big[0] & 0x7F) == 0x7F).sum()) == 0 and float((_e4m3_decode(big[0].cpu()) * float(s0.scale)).abs().max()) <= 448.0 * float(s0.scale) * 1.001
    sites.update()
    # LayerNorm forward / backward
    xf = rnd(2053, 768, dtype=torch.float32, seed=302)
    gamma, beta = rnd(768, dtype=torch.float32, seed=303) + 1.0, rnd(768, dtype=torch.float32, seed=304)
    s1, s2 = sites.site(("ln", 1)), sites.site(("ln", 2))
    dy = rnd(2053, 768, scale=0.1, seed=305)
    res = rnd(2053, 768, dtype=torch.float32, scale=0.1, seed=306)
    for rep in range(2):
        y16, _, mean, rstd, q8 = K.layernorm_fwd(xf, gamma, beta, 1e-6, want_fp8=True, q8=s1)
        dg, db, deferred = torch.zeros(768, device=DEV), torch.zeros(768, device=DEV), []
        dcol = torch.full((768,), 0.5, device=DEV)  # the column sums of dx are ADDED (the bias gradient of the projection that produced x)
        dx32, dx16, dq8, col_done = K.layernorm_bwd(dy, xf, gamma, beta, mean, rstd, dx_residual=res, want_f32=True, want_bf16=True, dgamma=dg, dbeta=db,
                                                    deferred=deferred, q8=s2, q8_colsum=dcol)
        K.ln_param_reduce_batched(deferred)
        dg0, db0 = torch.zeros(768, device=DEV), torch.zeros(768, device=DEV)
        ref32, _ = K.layernorm_bwd(dy, xf, gamma, beta, mean, rstd, dx_residual=res, want_f32=True, want_bf16=True, dgamma=dg0, dbeta=db0)
        assert col_done and torch.equal(ref32, dx32)  # the third partial row changes nothing else
        want = 0.5 + dx32.double().sum(0)
        assert float((dcol.double() - want).abs().max()) <= 1e-5 * float(dx32.double().abs().sum(0).max())
        assert torch.allclose(dg, dg0, rtol=1e-5, atol=1e-5) and torch.allclose(db, db0, rtol=1e-5, atol=1e-5)
        if rep == 0:
            assert q8[1].numel() == 2053 and dq8 is None  # first step: per-row copy from the forward, no copy from the backward
            sites.update()
        else:
            assert q8[1].numel() == 1
            check(q8, y16, "LayerNorm forward")
            check(dq8, dx16, "LayerNorm backward")
    # GEMM epilogues
    a, w = rnd(2053, 512, scale=0.5, seed=307), rnd(1024, 512, scale=0.05, seed=308)
    bias = rnd(1024, dtype=torch.float32, seed=309)
    a8, sa = K.quantize_fp8_rows(a)
    w8, sw = K.quantize_fp8(w)
    gin = rnd(2053, 1024, seed=310)
    s3, s4, s5, s6 = (sites.site(("gemm", i)) for i in range(4))
    for rep in range(2):
        outs = []
        for site, fn in ((s3, lambda o8: K.gemm(a, w, bias=bias, act=1, out8=o8)), (s4, lambda o8: K.gemm_fp8(a8, sa, w8, sw, bias=bias, act=1, out8=o8)),
                         (s5, lambda o8: K.gemm(a, w, gelu_in=gin, gelu_deriv=True, out8=o8)), (s6, lambda o8: K.gemm_fp8(a8, sa, w8, sw, gelu_in=gin, gelu_deriv=True, out8=o8))):
            o8 = torch.empty(2053, 1024, dtype=torch.uint8, device=DEV) if site.ready else None
            outs.append((site, fn((site, o8)), o8))
        if rep == 0:
            sites.update()
        else:
            for i, (site, y16, o8) in enumerate(outs):
                check((o8, site.scale), y16, f"GEMM epilogue {i}")


def test_q8_column_sums_strip_sums_and_8bit_only_outputs() -> None:
    """The pieces around the e4m3 weight-gradient path that carry the BIAS gradients and drop unread bf16 tensors: (1) ``cinema_quantize_fp8_site_colsum`` = the
    stand-alone 8-bit copy + column sums in one pass (sums vs fp32 torch, accumulated into an existing gradient; the copy equals the plain stand-alone copy bit
    for bit); (2) ``colsum_partials`` of the GEMM epilogues: sums over strips of 32 rows, every element written once - summed over the strips they equal the column
    sums of the bf16 output (ragged row count, both the BK = 64 and BK = 32 kernels, bf16 and e4m3 operands); (3) ``skip_d``: only the 8-bit copy is written, and it
    equals the copy of the launch that also wrote the bf16 tensor; (4) ``cinema_dequantize_fp8`` inverts the copy to within e4m3 rounding."""
    from cinema_amd import tape as T

    sites = T.Fp8Sites(torch.device(DEV, torch.cuda.current_device()))
    x = rnd(5003, 768, scale=0.7, seed=401)
    s0 = sites.site(("cs", 0))
    base = rnd(768, dtype=torch.float32, seed=402)
    acc = base.clone()
    assert K.quantize_fp8_site_colsum(x, s0, acc) is None  # first launch: records the maximum, still sums
    close(acc, base + x.float().sum(0), 1e-5, 1e-3 * float(x.float().sum(0).abs().max()), "column sums (first launch)")
    sites.update()
    acc2 = torch.zeros(768, device=DEV)
    q_a = K.quantize_fp8_site_colsum(x, s0, acc2)
    q_b = K.quantize_fp8_site(x, s0)
    assert torch.equal(q_a[0], q_b[0]) and float(q_a[1]) == float(q_b[1])
    close(acc2, x.float().sum(0), 1e-5, 1e-3 * float(x.float().sum(0).abs().max()), "column sums")
    back = K.dequantize_fp8(q_b)
    assert float((back.float() - x.float()).abs().max()) <= float(x.float().abs().max()) * 2.0 ** -4 + float(q_b[1]) * 2.0 ** -9
    # strip sums + 8-bit-only output of the GEMM epilogues
    for k in (768, 512):  # BK = 64 kernel / BK = 32 kernel
        m, n = 2053, 1024
        a, w = rnd(m, k, scale=0.5, seed=403), rnd(n, k, scale=0.05, seed=404)
        gin = rnd(m, n, seed=405)
        a8, sa = K.quantize_fp8_rows(a)
        w8, sw = K.quantize_fp8(w)
        site = sites.site(("strips", k))
        K.gemm(a, w, gelu_in=gin, gelu_deriv=True, out8=(site, None))
        sites.update()
        for name, fn in (("bf16", lambda **kw: K.gemm(a, w, gelu_in=gin, gelu_deriv=True, **kw)), ("e4m3", lambda **kw: K.gemm_fp8(a8, sa, w8, sw, gelu_in=gin, gelu_deriv=True, **kw))):
            strips = torch.full(((m + 31) // 32, n), float("nan"), device=DEV)
            o8 = torch.empty(m, n, dtype=torch.uint8, device=DEV)
            y = fn(out8=(site, o8), colsum_partials=strips)
            assert bool(torch.isfinite(strips).all()), "every strip element is written"
            ref = y.float().sum(0)
            close(strips.sum(0), ref, 1e-3, 4e-3 * float(ref.abs().max()), f"strip sums {name} k={k}")  # (the strips sum the fp32 values before the bf16 rounding)
            want = y.float().view(-1)[: 32 * n].view(32, n).sum(0)
            close(strips[0], want, 1e-3, 4e-3 * float(want.abs().max()), f"first strip {name} k={k}")
            if name == "e4m3":
                o8b = torch.empty(m, n, dtype=torch.uint8, device=DEV)
                strips_b = torch.empty_like(strips)
                assert K.gemm_fp8(a8, sa, w8, sw, gelu_in=gin, gelu_deriv=True, out8=(site, o8b), colsum_partials=strips_b, skip_d=True) is None
                assert torch.equal(o8b, o8) and torch.equal(strips_b, strips)


def test_adamw_groups_one_launch_equals_one_launch_per_group() -> None:
    """cinema_adamw_groups (the layer-decay parameter groups of a fine-tuning step in ONE launch) against one cinema_adamw launch per group: bit-identical
    parameters, moments and bf16 shadows over three steps; gaps between the ranges are left untouched; a non-finite norm skips every group."""
    torch.manual_seed(0)
    n = 40000
    bounds = [(0, 4096, 1e-3, 0.0), (4096, 12288, 5e-4, 0.05), (12288, 12296, 2e-3, 0.05), (16384, 40000, 1e-4, 0.01)]  # a gap [12296, 16384)
    g = torch.randn(n, device=DEV)
    state = {}
    for form in ("grouped", "single"):
        p, m, v = torch.linspace(-1, 1, n, device=DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        sh = p.bfloat16()
        coef, norm, st = torch.ones(1, device=DEV), torch.zeros(1, device=DEV), torch.zeros(2, dtype=torch.int32, device=DEV)
        for step in range(3):
            K.clip_coef((g * g).sum().reshape(1) if step != 1 else torch.tensor([float("nan")], device=DEV), 5.0, coef, norm, st)
            if form == "grouped":
                K.adamw_groups(p, g, m, v, bounds, 0.9, 0.95, 1e-8, coef, sh, st)
            else:
                for a, b, lr, wd in bounds:
                    K.adamw(p[a:b], g[a:b], m[a:b], v[a:b], lr, 0.9, 0.95, 1e-8, wd, 1, clip=coef, shadow=sh[a:b], step_state=st)
        state[form] = (p, m, v, sh, st.tolist())
    for a, b in zip(state["grouped"][:4], state["single"][:4]):
        assert torch.equal(a, b)
    assert state["grouped"][4] == state["single"][4] == [2, 1]  # two applied, the NaN one skipped
    assert torch.equal(state["grouped"][0][12296:16384], torch.linspace(-1, 1, n, device=DEV)[12296:16384])


def test_shared_weight_gradient_launches_stay_on_one_stream(monkeypatch: pytest.MonkeyPatch) -> None:
    """A weight used several times per step (dec_linear: the reference applies one nn.Linear to every view, cinema/mae/mae.py:486-493) gets one ACCUMULATING
    weight-gradient launch per use.  With two weight-gradient streams those must follow each other on one stream; dealt alternately they ran at the same time
    and lost updates (round 5, found by tests/test_ddp_gpu.py: dec_linear.weight's gradient 20 % off in ~1 run of 8).  Six uses of one buffer interleaved with
    launches for other buffers, against the fp32 sum.  (Not bit-compared: the split-K reduce of a small output adds its slices with fp32 atomics, csrc/gemm.hip
    splitk_reduce_body, so a non-zero destination is reproducible to 1 ulp only.)"""
    from cinema_amd import tape as T  # noqa: N812

    torch.manual_seed(0)
    m, n, k = 16384, 256, 768
    dys = [(torch.randn(m, n, device=DEV) * 0.1).bfloat16() for _ in range(6)]
    xs = [(torch.randn(m, k, device=DEV) * 0.1).bfloat16() for _ in range(6)]
    ref = sum(d.float().t() @ x.float() for d, x in zip(dys, xs))
    bref = sum(d.float().sum(0) for d in dys)
    for streams in (1, 2, 2, 2, 2, 2):
        monkeypatch.setattr(T, "SIDE_STREAMS", streams)
        shared, bias = torch.zeros(n, k, device=DEV), torch.zeros(n, device=DEV)
        others = [torch.zeros(n, k, device=DEV) for _ in range(6)]
        torch.cuda.synchronize()
        for i in range(6):
            T._wgrad_single(dys[i], xs[i], shared, bias)               # noqa: SLF001
            T._wgrad_single(dys[i], xs[(i + 1) % 6], others[i], None)  # noqa: SLF001  (keeps the alternation going)
        if streams == 2:
            assert len(set(T._DST_STREAM.values())) == 2 and len(T._DST_STREAM) == 8  # noqa: SLF001  (both streams in use; shared + bias + six others)
        T.join_side_stream(release=True)
        torch.cuda.synchronize()
        assert float((shared - ref).norm() / ref.norm()) < 1e-4, streams
        assert float((bias - bref).norm() / bref.norm()) < 1e-4, streams
        for i, o in enumerate(others):
            r = dys[i].float().t() @ xs[(i + 1) % 6].float()
            assert float((o - r).norm() / r.norm()) < 1e-4
    assert not T._DST_STREAM  # noqa: SLF001  (cleared with the join at the end of a backward pass)
