"""Tape ops of the transformer blocks: LayerNorm, Linear, MLP (GELU fused), self / cross attention (fused q|k|v projections, shared decoder k|v), dropout / drop-path, casts.

Part of the tape (``cinema_amd/tape/__init__.py`` holds :class:`Tape`, :class:`Var`, the weight caches, the weight-gradient streams and groups, the fp8 sites and the
autograd bridge); everything here is re-exported there, so callers keep writing ``tape.op_*``.  Module-level switches live in the package and are read through it
(``T.<NAME>``) so that an assignment ``tape.<NAME> = ...`` is seen here."""
from __future__ import annotations

import math
from typing import Callable

import torch

from cinema_amd import hip as K
from cinema_amd import tape as T
from cinema_amd.tape import (  # noqa: F401
    BF16, F32, Tape, Var, _adjacent, _fp8_ok, _tensor_scaled, a_fp8, b_cat, dgrad, flush_wgrads, fp8_site, mark_params, w_cat, w_fp8, w_plain, wgrad,
    wgrad8_problem, wgrad_problem,
)

__all__ = ['SharedKV', '_op_thin_linear', 'begin_stochastic', 'next_salt', 'op_cast_bf16', 'op_cast_f32', 'op_cross_attention', 'op_dropout', 'op_droppath_add', 'op_layernorm', 'op_linear', 'op_mlp', 'op_self_attention', 'op_shared_kv', 'share_kv_ok']


# --------------------------------------------------------------------------------------------------------------
# ops
# --------------------------------------------------------------------------------------------------------------
def op_layernorm(tape: Tape, x: Var, gamma: torch.nn.Parameter, beta: torch.nn.Parameter, eps: float, *, act: int = 0, out_f32: bool = False,
                 fp8: bool = False) -> Var:
    """y = [gelu](LN(x)); x fp32/bf16 [rows, c]; output bf16 (GEMM operand) or fp32 (residual stream)."""
    if fp8 and not out_f32 and T.FP8_FORWARD and x.data.is_cuda and x.data.shape[1] % 16 == 0:
        y16, y32, mean, rstd, q8 = K.layernorm_fwd(x.data, gamma.detach(), beta.detach(), eps, act=act, want_bf16=True, want_f32=False, want_fp8=True,
                                                   q8=fp8_site(x.data, gamma, "ln_out"))
        y = Var(y16)
        y.fp8 = q8
    else:
        y16, y32, mean, rstd = K.layernorm_fwd(x.data, gamma.detach(), beta.detach(), eps, act=act, want_bf16=not out_f32, want_f32=out_f32)
        y = Var(y32 if out_f32 else y16)
    gv, bv = tape.pvar(gamma), tape.pvar(beta)

    def bwd() -> None:
        if y.grad is None:
            return
        c = x.data.shape[1]
        want16 = x.data.dtype == F32  # fp32 residual-stream input: also emit the bf16 copy for the upstream GEMMs
        res = x.grad if (x.grad is not None and x.grad.dtype == F32) else None
        site8 = x.grad8_site if (x.data.dtype == F32 and T.FP8_WGRAD and T.FP8_FORWARD) else None  # the producer of x wants an 8-bit copy of the complete gradient
        if site8 is not None and site8.ready and T.FP8_DGRAD:
            want16 = False  # the e4m3 copy is what the upstream data- / weight-gradient GEMMs read; a bf16 reader (fallback) casts the fp32 rows lazily
        bias_p = x.grad8_bias if site8 is not None else None
        colsum = tape.pvar(bias_p).grad_buffer((c,)) if (bias_p is not None and bias_p.requires_grad) else None
        out = K.layernorm_bwd(y.grad, x.data, gamma.detach(), beta.detach(), mean, rstd, act=act, dx_residual=res,
                              want_f32=x.data.dtype == F32, want_bf16=want16 or x.data.dtype == BF16,
                              dgamma=gv.grad_buffer((c,)), dbeta=bv.grad_buffer((c,)), deferred=tape.pending_ln, q8=site8,
                              **({"q8_colsum": colsum} if site8 is not None else {}))
        dx32, dx16 = out[0], out[1]
        if x.data.dtype == F32:
            x.grad, x.grad16 = dx32, dx16  # includes the previously accumulated residual gradient
            x.grad8 = out[2] if site8 is not None else None
            x.grad8_bias_done = bool(site8 is not None and colsum is not None and out[3])
        else:
            x.add_grad(dx16)

    tape.record(bwd)
    return y


def op_linear(tape: Tape, x: Var, weight: torch.nn.Parameter, bias: torch.nn.Parameter | None, *, residual: Var | None = None,
              out_f32: bool = False, row_mask: torch.Tensor | None = None, w16: torch.Tensor | None = None,
              to_param_layout: Callable | None = None, fp8: bool = False) -> Var:
    """y = x W^T + b (+ residual); x bf16 [m,k]; W given as nn.Linear / 1x1-conv weight (or a pre-built shadow ``w16``)."""
    if (w16 is None and out_f32 and residual is None and row_mask is None and weight.dim() >= 2 and weight.shape[0] < 8 and x.data.is_cuda
            and x.data.dtype == BF16 and x.data.is_contiguous() and math.prod(weight.shape[1:]) == x.data.shape[1] <= 64 and x.data.shape[1] % 8 == 0):
        return _op_thin_linear(tape, x, weight, bias)
    if (w16 is None and out_f32 and residual is None and row_mask is None and weight.dim() >= 2 and x.data.is_cuda and x.data.dtype == BF16 and x.data.is_contiguous()
            and math.prod(weight.shape[1:]) == x.data.shape[1] and K.fanout_ok(weight.shape[0], x.data.shape[1])):
        return _op_thin_linear(tape, x, weight, bias, fanout=True)  # few inputs, a few dozen outputs (the 1 -> 32 channel shortcut of the raw-image block)
    w = w16 if w16 is not None else w_plain(weight)
    x8t = None  # per-tensor e4m3 copy of x (also the X operand of the weight gradient)
    site_dy = None
    if fp8 and w16 is None and row_mask is None and _fp8_ok(x.data, weight) and (residual is None or residual.data.dtype == F32):
        site_x = fp8_site(x.data, weight, "x")
        if site_x is not None:
            if _tensor_scaled(x.fp8):
                x8t = x.fp8
            else:
                x8t = K.quantize_fp8_site(x.data, site_x)  # None in the site's first step (records the maximum)
            site_dy = fp8_site(x.data, weight, "dy")
        x8, sx = x8t if x8t is not None else a_fp8(x)
        w8, sw = w_fp8(weight)
        y = Var(K.gemm_fp8(x8, sx, w8, sw, bias=None if bias is None else bias.detach(), residual=None if residual is None else residual.data,
                           out_dtype=F32 if (out_f32 or residual is not None) else BF16))
        if y.data.dtype == F32:
            y.grad8_site = site_dy  # the LayerNorm backward that completes this residual-stream gradient writes its e4m3 copy
            y.grad8_bias = bias if site_dy is not None else None  # ... and sums its columns into this bias' gradient
    else:
        y = Var(K.gemm(x.data, w, bias=None if bias is None else bias.detach(), residual=None if residual is None else residual.data,
                       out_dtype=F32 if (out_f32 or residual is not None) else BF16, row_mask=row_mask))
    wv, bv = tape.pvar(weight), tape.pvar(bias)

    def bwd() -> None:
        if y.grad is None:
            return
        if residual is not None:
            residual.add_grad(y.grad, y.grad16)
        # row_mask contract: y = mask * (xW^T + b) and the consumer (op_dwconv with in_mask) hands back a gradient whose
        # masked rows are already zero, so the weight/bias gradients below need no extra masking pass.
        g8 = y.grad8 if (site_dy is not None and _tensor_scaled(y.grad8)) else None
        # with the e4m3 copy of the gradient in hand nobody may need its bf16 form (the LayerNorm backward then did not write one): ask for it only on the bf16 paths
        fp8_dg = g8 is not None and fp8 and w16 is None and row_mask is None and T.FP8_DGRAD and T.w_fp8_t(weight) is not None and weight.shape[0] % 16 == 0 and weight.shape[1] % 8 == 0
        fp8_wg = g8 is not None and x8t is not None and to_param_layout is None and weight.shape[0] % 16 == 0 and x.data.shape[1] % 16 == 0
        want_bias = bias is not None and bias.requires_grad and not y.grad8_bias_done  # (done: the LayerNorm backward that produced the gradient summed its columns)
        if weight.requires_grad:
            if fp8_wg:
                wgrad8_problem(tape, g8, x8t, wv.grad_buffer(tuple(w.shape)).view(-1, x.data.shape[1]), y.grad16 if y.grad16 is not None else y.grad,
                               bv.grad_buffer((weight.shape[0],)) if want_bias else None)
            else:
                wgrad(tape, y.grad_bf16(), x.data, wv, bv if want_bias else None, tuple(w.shape), to_param_layout)
        if x.needs_grad:
            x.add_grad(dgrad(None if (fp8_dg and y.grad.dtype != BF16) else y.grad_bf16(), weight, w, row_mask=row_mask, fp8=fp8 and w16 is None, dy8=g8))

    tape.record(bwd)
    return y


def _op_thin_linear(tape: Tape, x: Var, weight: torch.nn.Parameter, bias: torch.nn.Parameter | None, fanout: bool = False) -> Var:
    """A head with fewer than 8 outputs (the 4-class segmentation head over every voxel) or, ``fanout``, a layer with at most 8 inputs: streaming
    kernels on the fp32 master weight."""
    w2 = weight.detach().reshape(weight.shape[0], -1)
    fwd_k, bwd_k = (K.fanout_linear_fwd, K.fanout_linear_bwd) if fanout else (K.thin_linear_fwd, K.thin_linear_bwd)
    y = Var(fwd_k(x.data, w2, None if bias is None else bias.detach()))
    wv, bv = tape.pvar(weight), tape.pvar(bias)

    def bwd() -> None:
        if y.grad is None:
            return
        dw = wv.grad_buffer(tuple(w2.shape)) if weight.requires_grad else None
        db = bv.grad_buffer((w2.shape[0],)) if (bias is not None and bias.requires_grad) else None
        dx = bwd_k(x.data, w2, y.grad.contiguous(), dw, db, want_dx=x.needs_grad)
        if dx is not None:
            x.add_grad(dx)

    tape.record(bwd)
    return y


# fc1's epilogue evaluates the erf terms of the GELU anyway: with this flag it stores GELU'(pre-activation) (bf16) instead of the pre-activation, and the
# data gradient through the activation (fc2's dgrad epilogue) is one multiply per element instead of a second erf evaluation (that epilogue was VALU-bound:
# 10960x3072x768 data gradient 93.7 us with the erf against 61.5 us plain).


def op_mlp(tape: Tape, x: Var, fc1_w: torch.nn.Parameter, fc1_b: torch.nn.Parameter, fc2_w: torch.nn.Parameter, fc2_b: torch.nn.Parameter,
           residual: Var | None, fp8: bool = False) -> Var:
    """[residual +] fc2(gelu(fc1(x))) (timm Mlp / ConvMlp); GELU forward fused into fc1's epilogue, GELU backward into fc2's dgrad.
    ``residual=None``: the caller adds it (behind a DropPath, :func:`op_droppath_add`)."""
    w1, w2 = w_plain(fc1_w), w_plain(fc2_w)
    m, hidden = x.data.shape[0], w1.shape[0]
    deriv = True
    # h holds GELU'(fc1 output), not the fc1 output.  (The library can also hold GELU' as an 8-bit code - csrc/common.cuh gelu8_*, a uint8 auxiliary tensor in
    # hip.gemm - which halves these bytes; measured neutral on the step in round 5, profiles/r05_g_gelu8_ab.txt, so the step does not use it.)
    h = K.empty((m, hidden), dtype=BF16, device=x.data.device)
    x8t = a8t = None  # per-tensor e4m3 copies of x (LayerNorm output) and of the GELU output: operands of the forward AND weight-gradient GEMMs
    site_dy = site_dh = None
    if fp8 and _fp8_ok(x.data, fc1_w, fc2_w) and (residual is None or residual.data.dtype == F32):
        site_a = fp8_site(x.data, fc2_w, "x")
        if site_a is not None:
            site_dy, site_dh = fp8_site(x.data, fc2_w, "dy"), fp8_site(x.data, fc1_w, "dy")
            x8t = x.fp8 if _tensor_scaled(x.fp8) else None
            a8 = K.empty((m, hidden), dtype=torch.uint8, device=x.data.device) if site_a.ready else None
        x8, sx = a_fp8(x)
        # with its e4m3 copy in hand the bf16 GELU output has no reader in the steady state (fc2's forward and weight gradient take the copy): it is not
        # written at all (113 MB per ViT-Large block); a consumer outside the e4m3 GEMMs (fallback paths below) gets it by one dequantisation pass
        a8_only = site_a is not None and a8 is not None
        a = K.gemm_fp8(x8, sx, *w_fp8(fc1_w), bias=fc1_b.detach(), act=1, aux_out=h, gelu_deriv=deriv, out8=None if site_a is None else (site_a, a8), skip_d=a8_only)
        if site_a is not None and a8 is not None:
            a8t = (a8, site_a.scale)
        a8r, sa = a8t if a8t is not None else K.quantize_fp8_rows(a)
        y = Var(K.gemm_fp8(a8r, sa, *w_fp8(fc2_w), bias=fc2_b.detach(), residual=None if residual is None else residual.data, out_dtype=F32))
        y.grad8_site = site_dy
        y.grad8_bias = fc2_b if site_dy is not None else None
    else:
        a = K.gemm(x.data, w1, bias=fc1_b.detach(), act=1, aux_out=h, gelu_deriv=deriv)
        y = Var(K.gemm(a, w2, bias=fc2_b.detach(), residual=None if residual is None else residual.data, out_dtype=F32))
    pv = [tape.pvar(p) for p in (fc1_w, fc1_b, fc2_w, fc2_b)]
    a_box = [a]

    def bwd() -> None:
        if y.grad is None:
            return
        if residual is not None:
            residual.add_grad(y.grad, y.grad16)
        g8 = y.grad8 if (site_dy is not None and _tensor_scaled(y.grad8)) else None
        ok8 = hidden % 16 == 0 and x.data.shape[1] % 16 == 0
        fp8_dg2 = g8 is not None and T.w_fp8_t(fc2_w) is not None and T.FP8_DGRAD  # fc2's data gradient reads the e4m3 copy of the gradient
        if g8 is not None and a8t is not None and ok8:
            # bias gradient = column sums of the gradient: of its bf16 copy when one exists, of the fp32 rows otherwise
            wgrad8_problem(tape, g8, a8t, pv[2].grad_buffer(tuple(w2.shape)), y.grad16 if y.grad16 is not None else y.grad,
                           None if y.grad8_bias_done else pv[3].grad_buffer((w2.shape[0],)))
        else:
            if a_box[0] is None:
                a_box[0] = K.dequantize_fp8(a8t)
            wgrad(tape, y.grad_bf16(), a_box[0], pv[2], None if y.grad8_bias_done else pv[3], tuple(w2.shape))
        dy16 = y.grad16 if (fp8_dg2 and y.grad.dtype != BF16) else y.grad_bf16()  # may be None: nobody reads it on the e4m3 path
        dh8 = K.empty((m, hidden), dtype=torch.uint8, device=x.data.device) if (site_dh is not None and site_dh.ready) else None
        # fc1's bias gradient = column sums of dh: the data-gradient epilogue that produces dh leaves its sums per strip of 32 rows (3.5 MB), one small launch adds them up
        strips = K.empty(((m + 31) // 32, hidden), dtype=F32, device=x.data.device) if (site_dh is not None and ok8 and not K.FORCE_GENERIC and T.w_fp8_t(fc2_w) is not None) else None
        # dh in bf16 has no reader either when fc1's weight AND data gradient take the e4m3 copy and the strips give the bias gradient
        dh8_only = (fp8_dg2 and dh8 is not None and x8t is not None and ok8 and strips is not None and T.w_fp8_t(fc1_w) is not None
                    and fc1_w.shape[0] % 16 == 0 and fc1_w.shape[1] % 8 == 0)
        if dh8_only:
            wt2 = T.w_fp8_t(fc2_w)
            K.gemm_fp8(g8[0], g8[1], wt2[0], wt2[1], gelu_in=h, gelu_deriv=deriv, out8=(site_dh, dh8), colsum_partials=strips, skip_d=True)
            dh = None
        else:
            dh = dgrad(dy16 if dy16 is not None else y.grad_bf16(), fc2_w, w2, gelu_in=h, fp8=fp8, gelu_deriv=deriv, dy8=g8,
                       out8=None if site_dh is None else (site_dh, dh8), colsum_partials=strips)
        if strips is not None:
            K.colsum(strips, pv[1].grad_buffer((w1.shape[0],)))
        dh8t = None if dh8 is None else (dh8, site_dh.scale)
        if dh8t is not None and x8t is not None and ok8:
            wgrad8_problem(tape, dh8t, x8t, pv[0].grad_buffer(tuple(w1.shape)), dh, None if strips is not None else pv[1].grad_buffer((w1.shape[0],)))
        else:
            wgrad(tape, dh, x.data, pv[0], None if strips is not None else pv[1], tuple(w1.shape))
        if x.needs_grad:
            x.add_grad(dgrad(dh, fc1_w, w1, fp8=fp8, dy8=dh8t))

    tape.record(bwd)
    return y


def op_self_attention(tape: Tape, x: Var, batch: int, heads: int, q_w, q_b, kv_w, kv_b, rope: tuple | None = None, fp8: bool = False) -> Var:  # noqa: ANN001
    """Fused q|k|v projection (one N=3C GEMM on concatenated shadow weights) + flash attention.  x bf16 [b*t, c].
    ``rope`` = (cos, sin) fp32 [heads, hd/2]: the reference's head-indexed rotary embedding (``cinema/vit.py:496-499``), applied in place to the
    q|k columns of the projection; the backward pass rotates dq|dk back before the weight / data gradients."""
    c = x.data.shape[1]
    w = w_cat((q_w, kv_w))
    bias = b_cat((q_b, kv_b)) if q_b is not None else None
    if fp8 and _fp8_ok(x.data, q_w, kv_w):  # per-tensor weight scales: q and kv are two GEMMs into the column blocks of one buffer
        x8, sx = a_fp8(x)
        qkv = K.empty((x.data.shape[0], 3 * c), dtype=BF16, device=x.data.device)
        K.gemm_fp8(x8, sx, *w_fp8(q_w), bias=None if q_b is None else q_b.detach(), out=qkv[:, :c])
        K.gemm_fp8(x8, sx, *w_fp8(kv_w), bias=None if kv_b is None else kv_b.detach(), out=qkv[:, c:])
    else:
        qkv = K.gemm(x.data, w, bias=bias)
    if rope is not None:
        K.rope_heads(qkv, 2 * heads, heads, c // heads, rope[0], rope[1])
    t = qkv.shape[0] // batch
    q3 = qkv.view(batch, t, 3 * c)
    scale = (c // heads) ** -0.5
    o, lse, o_lo = K.attention_fwd(q3[..., :c], q3[..., c:2 * c], q3[..., 2 * c:], heads, scale, want_lo=True) if (tape.train and T.ATTN_O_LO) else \
        (*K.attention_fwd(q3[..., :c], q3[..., c:2 * c], q3[..., 2 * c:], heads, scale), None)
    y = Var(o.view(batch * t, c))
    pv = [tape.pvar(p) for p in (q_w, q_b, kv_w, kv_b)]

    def bwd() -> None:
        if y.grad is None:
            return
        dqkv = K.empty_like(qkv)
        d3 = dqkv.view(batch, t, 3 * c)
        K.attention_bwd(q3[..., :c], q3[..., c:2 * c], q3[..., 2 * c:], o, y.grad.view(batch, t, c), lse, heads, scale, d3[..., :c],
                        d3[..., c:2 * c], d3[..., 2 * c:], o_lo=o_lo)
        if T.ATTN_CAPTURE is not None:  # dev tooling (tools/attn_dq_error.py): the operands of this block's attention backward
            T.ATTN_CAPTURE.append(dict(qkv=qkv.clone(), o=o.clone(), do=y.grad.clone(), lse=lse.clone(), dqkv=dqkv.clone(), x=x.data.clone(), batch=batch, heads=heads))
        if rope is not None:
            K.rope_heads(dqkv, 2 * heads, heads, c // heads, rope[0], rope[1], inverse=True)
        gq, gkv = pv[0].grad_buffer((c, c)), pv[2].grad_buffer((2 * c, c))
        g3 = b3 = None
        if q_b is not None and pv[0].direct and pv[2].direct and _adjacent(gq, gkv):
            bq, bkv = pv[1].grad_buffer((c,)), pv[3].grad_buffer((2 * c,))
            if pv[1].direct and pv[3].direct and _adjacent(bq, bkv):  # one [3c, c] weight-gradient GEMM + one column sum
                g3, b3 = gq.as_strided((3 * c, c), (c, 1)), bq.as_strided((3 * c,), (1,))
        # e4m3 weight gradient (T.FP8_WGRAD): dY = one stand-alone 8-bit copy of dq|dk|dv under a delayed per-tensor scale (the same pass sums its columns =
        # the bias gradient), X = the LayerNorm's per-tensor copy.  The site is created and fed in EVERY step, whatever the state of the other sites: a
        # site that first appears a step late would miss the recording
        site = fp8_site(dqkv, q_w, "dy") if (fp8 and c % 16 == 0) else None
        d8, bias_done = None, False
        if site is not None:
            if b3 is not None:
                d8, bias_done = K.quantize_fp8_site_colsum(dqkv, site, b3), True
            else:
                d8 = K.quantize_fp8_site(dqkv, site)
            if not _tensor_scaled(x.fp8):
                d8 = None
        if g3 is not None:
            if d8 is not None:
                wgrad8_problem(tape, d8, x.fp8, g3, dqkv, None if bias_done else b3)
            else:
                wgrad_problem(tape, dqkv, x.data, g3, None if bias_done else b3)
        elif d8 is not None:
            wgrad8_problem(tape, (d8[0][:, :c], d8[1]), x.fp8, gq, dqkv[:, :c], None if q_b is None else pv[1].grad_buffer((c,)))
            wgrad8_problem(tape, (d8[0][:, c:], d8[1]), x.fp8, gkv, dqkv[:, c:], None if kv_b is None else pv[3].grad_buffer((2 * c,)))
        else:
            wgrad(tape, dqkv[:, :c], x.data, pv[0], pv[1], (c, c))
            wgrad(tape, dqkv[:, c:], x.data, pv[2], pv[3], (2 * c, c))
        if x.needs_grad:
            # data gradient of the fused projection: one e4m3 GEMM over K = 3c on the 8-bit copy of dq|dk|dv and a JOINT transposed shadow of [W_q; W_kv]
            # (one scale for the pair), else the bf16 GEMM on the concatenated shadow
            wt = None
            if site is not None and T.FP8_DGRAD:  # (asked for in EVERY step of this path, the first included: its descriptors must exist before a step is recorded)
                flat = getattr(q_w, "_cinema_flat", None)
                wt = flat.fp8_shadow_cat_t((q_w, kv_w)) if flat is not None else None
            if wt is not None and d8 is not None:
                x.add_grad(K.gemm_fp8(d8[0], d8[1], wt[0], wt[1]))
            else:
                x.add_grad(K.gemm(dqkv, w, a_kmajor=True, b_kmajor=False))

    tape.record(bwd)
    return y


class SharedKV:
    """k|v projections of ALL decoder blocks from one GEMM (:func:`op_shared_kv`): ``data`` [b*tk, n_blocks * 2c] bf16, ``grad`` the same shape,
    filled block by block in the backward pass and consumed by one data-gradient GEMM + one grouped weight-gradient launch."""

    def __init__(self, data: torch.Tensor, width: int) -> None:
        self.data, self.width, self.grad = data, width, None

    def part(self, i: int) -> torch.Tensor:
        return self.data[:, i * self.width:(i + 1) * self.width]

    def grad_part(self, i: int) -> torch.Tensor:
        if self.grad is None:
            self.grad = K.zeros(tuple(self.data.shape), self.data.dtype, self.data.device)  # zero: a block whose backward returns early leaves its column block untouched
        return self.grad[:, i * self.width:(i + 1) * self.width]


# The decoder blocks all project the SAME un-normed encoder output to their keys / values (reference cinema/mae/mae.py:580-582, cinema/vit.py:472-477):
# one GEMM with the concatenated weights (N = n_blocks * 2c: 10.75 rounds of tiles instead of 8 x 1.34), one data-gradient GEMM with K = n_blocks * 2c
# and one grouped weight-gradient launch instead of 8 of each: 423 vs 610 us per step measured in isolation (tools/bench_gemm.py "X dec").  Off with fp8
# forward.  Under a gradient exchange the k|v parameters form a marked range of their own (their gradients are complete after the shared backward).


def share_kv_ok(xk: Var, attns: list) -> bool:
    # (legal under a gradient exchange since round 4: the k|v parameters are taken out of their blocks' marked ranges and marked as a range of their own,
    # which fires after the shared backward below - see op_shared_kv / Block.tape_forward)
    return (len(attns) > 1 and not T.FP8_FORWARD and xk.data.is_cuda and not K.FORCE_GENERIC
            and all(a.kv.weight.shape == attns[0].kv.weight.shape and (a.kv.bias is None) == (attns[0].kv.bias is None) for a in attns)
            and attns[0].kv.weight.shape[0] % 16 == 0 and xk.data.shape[1] % 8 == 0)


def op_shared_kv(tape: Tape, xk: Var, attns: list) -> SharedKV:
    """k|v of every block in ``attns`` (modules with ``kv`` Linear layers) from xk bf16 [b*tk, c]."""
    n, (two_c, c) = len(attns), attns[0].kv.weight.shape
    rows = xk.data.shape[0]
    # gradient exchange: the k|v weight gradients of ALL blocks are complete only after this op's backward (recorded first = runs last): their ranges are
    # marked here, the blocks leave them out of their own marks
    mark_params(tape, [p for a in attns for p in (a.kv.weight, a.kv.bias) if p is not None and p.requires_grad])
    wcat = K.empty((n * two_c, c), dtype=BF16, device=xk.data.device)
    K.row_copy_multi([dict(dst=wcat[i * two_c:(i + 1) * two_c], src=w_plain(a.kv.weight)) for i, a in enumerate(attns)])
    bcat = None
    if attns[0].kv.bias is not None:
        bcat = K.empty((n * two_c,), dtype=F32, device=xk.data.device)
        K.row_copy_multi([dict(dst=bcat[i * two_c:(i + 1) * two_c].view(1, two_c), src=a.kv.bias.detach().view(1, two_c)) for i, a in enumerate(attns)])
    shared = SharedKV(K.gemm(xk.data, wcat, bias=bcat), two_c)
    pvs = [(tape.pvar(a.kv.weight), tape.pvar(a.kv.bias)) for a in attns]

    def bwd() -> None:  # recorded before the blocks: runs after all of them in the backward pass
        if shared.grad is None:
            return
        prev, tape.grouping = tape.grouping, T.GROUP_WGRAD == 2
        for i, (a, (wv, bv)) in enumerate(zip(attns, pvs)):
            if a.kv.weight.requires_grad:
                wgrad(tape, shared.grad_part(i), xk.data, wv, bv if (a.kv.bias is not None and a.kv.bias.requires_grad) else None, (two_c, c))
        flush_wgrads(tape)
        tape.grouping = prev
        if xk.needs_grad:
            xk.add_grad(K.gemm(shared.grad, wcat, a_kmajor=True, b_kmajor=False))
        assert rows == shared.grad.shape[0]

    tape.record(bwd)
    return shared


def op_cross_attention(tape: Tape, xq: Var, xk: Var, batch: int, heads: int, q_w, q_b, kv_w, kv_b, fp8: bool = False, shared: tuple | None = None) -> Var:  # noqa: ANN001
    """q from xq (bf16 [b*tq, c]), k|v from xk (bf16 [b*tk, c], shared by every decoder block, not normed).  ``shared`` = (SharedKV, block index):
    the k|v projection (and its gradients) are handled by :func:`op_shared_kv` for all blocks at once."""
    c = xq.data.shape[1]
    wq, wkv = w_plain(q_w), w_plain(kv_w)
    fp8_sites_on = False
    if shared is not None:
        q = K.gemm(xq.data, wq, bias=None if q_b is None else q_b.detach())
        kv = shared[0].part(shared[1])
    elif fp8 and _fp8_ok(xq.data, q_w) and _fp8_ok(xk.data, kv_w):
        q = K.gemm_fp8(*a_fp8(xq), *w_fp8(q_w), bias=None if q_b is None else q_b.detach())
        if xk.fp8t is None and xk.fp8 is None:  # the keys are the same tensor for every decoder block: quantise once (the first block's site: per-tensor
            site_k = fp8_site(xk.data, kv_w, "x")  # delayed scale, the copy then also is the X operand of every block's e4m3 k|v weight gradient)
            if site_k is not None:
                xk.fp8t = K.quantize_fp8_site(xk.data, site_k)
            if xk.fp8t is None:
                xk.fp8 = K.quantize_fp8_rows(xk.data)
        kv = K.gemm_fp8(*(xk.fp8t if xk.fp8t is not None else xk.fp8), *w_fp8(kv_w), bias=None if kv_b is None else kv_b.detach())
        fp8_sites_on = True
    else:
        q = K.gemm(xq.data, wq, bias=None if q_b is None else q_b.detach())
        kv = K.gemm(xk.data, wkv, bias=None if kv_b is None else kv_b.detach())
    tq, tk = q.shape[0] // batch, kv.shape[0] // batch
    q3, kv3 = q.view(batch, tq, c), kv.view(batch, tk, 2 * c)
    scale = (c // heads) ** -0.5
    o, lse, o_lo = K.attention_fwd(q3, kv3[..., :c], kv3[..., c:], heads, scale, want_lo=True) if (tape.train and T.ATTN_O_LO) else \
        (*K.attention_fwd(q3, kv3[..., :c], kv3[..., c:], heads, scale), None)
    y = Var(o.view(batch * tq, c))
    pv = [tape.pvar(p) for p in (q_w, q_b, kv_w, kv_b)]

    def bwd() -> None:
        if y.grad is None:
            return
        dq = K.empty_like(q)
        dkv = K.empty_like(kv) if shared is None else shared[0].grad_part(shared[1])
        dkv3 = dkv.view(batch, tk, 2 * c)
        K.attention_bwd(q3, kv3[..., :c], kv3[..., c:], o, y.grad.view(batch, tq, c), lse, heads, scale, dq.view(batch, tq, c), dkv3[..., :c],
                        dkv3[..., c:], o_lo=o_lo)
        dq8 = dkv8 = None  # e4m3 copies of the gradients under delayed per-tensor scales: dY of the weight gradients, A of the data gradients
        if fp8_sites_on and c % 16 == 0:
            sq, skv = fp8_site(dq, q_w, "dy"), fp8_site(dq, kv_w, "dy")
            if sq is not None:  # one pass per gradient: 8-bit copy + column sums (= the bias gradient)
                dq8 = K.quantize_fp8_site_colsum(dq, sq, pv[1].grad_buffer((c,))) if q_b is not None else K.quantize_fp8_site(dq, sq)
                dkv8 = K.quantize_fp8_site_colsum(dkv, skv, pv[3].grad_buffer((2 * c,))) if kv_b is not None else K.quantize_fp8_site(dkv, skv)
        bias_done = fp8_sites_on and c % 16 == 0 and fp8_site(dq, q_w, "dy") is not None
        if dq8 is not None and _tensor_scaled(xq.fp8):
            wgrad8_problem(tape, dq8, xq.fp8, pv[0].grad_buffer((c, c)), dq, None)
        else:
            wgrad(tape, dq, xq.data, pv[0], None if bias_done else pv[1], (c, c))
        if shared is None:
            if dkv8 is not None and xk.fp8t is not None:
                wgrad8_problem(tape, dkv8, xk.fp8t, pv[2].grad_buffer((2 * c, c)), dkv, None)
            else:
                wgrad(tape, dkv, xk.data, pv[2], None if bias_done else pv[3], (2 * c, c))
        if xq.needs_grad:
            xq.add_grad(dgrad(dq, q_w, wq, fp8=fp8, dy8=dq8))
        if shared is None and xk.needs_grad:
            xk.add_grad(dgrad(dkv, kv_w, wkv, fp8=fp8, dy8=dkv8))

    tape.record(bwd)
    return y


def begin_stochastic(module: torch.nn.Module, device: torch.device) -> bool:
    """Call once at the start of a top-level forward: when ``module`` is in training mode and holds active dropout / drop-path layers, advance
    the device RNG step (one launch - new masks for this forward, also on every replay of a recorded step).  Returns whether it did."""
    active = module.__dict__.get("_cinema_stochastic")
    if active is None:
        active = module.__dict__["_cinema_stochastic"] = any(
            (isinstance(m, torch.nn.Dropout) and m.p > 0) or (type(m).__name__ == "DropPath" and getattr(m, "drop_prob", 0.0) > 0) for m in module.modules())
    if active and module.training:
        K.rng_advance(device)
        return True
    return False


def next_salt(tape: Tape) -> int:
    """Call-site id of a stochastic op inside one forward pass (the Philox counter stream of that op; deterministic in launch order)."""
    tape.salt = getattr(tape, "salt", 0) + 1
    return tape.salt


def op_dropout(tape: Tape, x: Var, p: float) -> Var:
    """``nn.Dropout(p)`` in training mode on bf16 rows (``cinema/conv.py:343``): y = x * keep / (1 - p); the backward pass regenerates the mask."""
    salt = next_salt(tape)
    y = Var(K.dropout(x.data, p, salt))

    def bwd() -> None:
        if y.grad is not None and x.needs_grad:
            x.add_grad(K.dropout(y.grad.contiguous(), p, salt))

    tape.record(bwd)
    return y


def op_droppath_add(tape: Tape, h: Var, residual: Var, batch: int, p: float) -> Var:
    """residual + DropPath(h) (timm ``DropPath`` as used at ``cinema/vit.py:606-609``): per-sample keep / (1 - p) factor on the fp32 rows of h
    ([batch * t, c], sample-major).  The factors are kept for the backward pass (batch floats)."""
    scale = K.droppath_scale(batch, p, next_salt(tape), h.data.device)
    rps = h.data.shape[0] // batch
    y = Var(K.scale_rows_add(h.data, scale, rps, residual=residual.data))

    def bwd() -> None:
        if y.grad is None:
            return
        residual.add_grad(y.grad, y.grad16)
        if h.needs_grad:
            if h.grad is None and h.data.is_cuda and h.data.shape[1] % 4 == 0:
                # the branch output's gradient has exactly two readers, the weight- and data-gradient GEMMs of the projection that produced it: written as bf16
                # (the fp32 tensor + a cast launch per projection cost 28 launches / 0.5 ms per ConvUNetR step)
                h.add_grad(K.scale_rows_bf16(y.grad.contiguous(), scale, rps))
            else:
                h.add_grad(K.scale_rows_add(y.grad.contiguous(), scale, rps))

    tape.record(bwd)
    return y


def op_cast_bf16(tape: Tape, x: Var) -> Var:
    y = Var(K.cast(x.data, BF16))

    def bwd() -> None:
        if y.grad is not None and x.needs_grad:
            if x.grad_any and x.grad is None:
                x.add_grad(y.grad)  # (the decoder's assembled keys: 5.6 M elements that were cast to fp32 only to be gathered and cast back)
            else:
                x.add_grad(K.cast(y.grad, F32))

    tape.record(bwd)
    return y


def op_cast_f32(tape: Tape, x: Var) -> Var:
    y = Var(K.cast(x.data, F32))

    def bwd() -> None:
        if y.grad is not None and x.needs_grad:
            x.add_grad(K.cast(y.grad, BF16))

    tape.record(bwd)
    return y
