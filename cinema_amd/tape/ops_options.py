"""Tape ops of the transformer-block OPTIONS no shipped CineMA config switches on but the reference constructors accept (``cinema/vit.py:446-609``): timm
``LayerScale`` (``init_values``), ``qk_norm`` (LayerNorm over the head dimension of q and k), ``proj_drop`` (``nn.Dropout`` behind the attention projection and inside
timm's ``Mlp``).  They run UNFUSED - separate projections, an attention op on given q / k / v, element-wise products - on the library's kernels: correct first, the
fused forms of ``ops_block`` stay the path of every configuration that is benchmarked.

Part of the tape (``cinema_amd/tape/__init__.py``); everything here is re-exported there."""
from __future__ import annotations

import torch

from cinema_amd import hip as K
from cinema_amd import tape as T
from cinema_amd.tape import BF16, F32, Tape, Var, dgrad, w_plain, wgrad  # noqa: F401

__all__ = ["op_add", "op_attention", "op_layerscale", "op_linear_gelu", "op_split_cols"]


def op_layerscale(tape: Tape, h: Var, gamma: torch.nn.Parameter) -> Var:
    """timm ``LayerScale``: y = h * gamma, gamma [c] (``cinema/vit.py:561,576``).  h fp32 or bf16 [m, c] -> fp32."""
    g = gamma.detach()
    hd = h.data.contiguous()
    y = Var(K.mul_rows(hd, g, F32))
    gv = tape.pvar(gamma)

    def bwd() -> None:
        if y.grad is None:
            return
        dy = y.grad.contiguous()
        if gamma.requires_grad:
            K.colsum(K.mul_rows(dy, hd, F32), gv.grad_buffer((g.numel(),)))  # d gamma[j] = sum_r dy[r][j] h[r][j]
        if h.needs_grad:
            h.add_grad(K.mul_rows(dy, g, hd.dtype))

    tape.record(bwd)
    return y


def op_add(tape: Tape, h: Var, residual: Var, batch: int) -> Var:
    """residual + h on fp32 rows (the residual add of a block whose branch output went through LayerScale / Dropout and therefore left the GEMM epilogue)."""
    ones = K.full((batch,), 1.0, F32, h.data.device)
    rps = h.data.shape[0] // batch
    y = Var(K.scale_rows_add(h.data.contiguous(), ones, rps, residual=residual.data))

    def bwd() -> None:
        if y.grad is None:
            return
        residual.add_grad(y.grad, y.grad16)
        if h.needs_grad:
            h.add_grad(y.grad)

    tape.record(bwd)
    return y


def op_linear_gelu(tape: Tape, x: Var, weight: torch.nn.Parameter, bias: torch.nn.Parameter | None) -> Var:
    """GELU(x W^T + b) as a tensor of its own (timm ``Mlp`` with ``drop`` > 0 puts ``nn.Dropout`` between the activation and fc2, so fc1 -> GELU -> fc2 cannot be
    one op): bf16 [m, hidden]; GELU'(pre-activation) is kept (bf16) for the backward pass."""
    w = w_plain(weight)
    m, hidden = x.data.shape[0], w.shape[0]
    deriv = K.empty((m, hidden), dtype=BF16, device=x.data.device)
    y = Var(K.gemm(x.data, w, bias=None if bias is None else bias.detach(), act=1, aux_out=deriv, gelu_deriv=True))
    wv, bv = tape.pvar(weight), tape.pvar(bias)

    def bwd() -> None:
        if y.grad is None:
            return
        dh = K.mul_rows(y.grad_bf16().contiguous(), deriv, BF16)
        if weight.requires_grad:
            wgrad(tape, dh, x.data, wv, bv if (bias is not None and bias.requires_grad) else None, tuple(w.shape))
        if x.needs_grad:
            x.add_grad(dgrad(dh, weight, w))

    tape.record(bwd)
    return y


def op_split_cols(tape: Tape, x: Var, widths: list) -> list:
    """Contiguous copies of consecutive column blocks of x [m, sum(widths)] (the k | v halves of the fused kv projection when k goes through ``k_norm``)."""
    m = x.data.shape[0]
    outs, off = [], 0
    for w in widths:
        t = K.empty((m, w), dtype=x.data.dtype, device=x.data.device)
        K.row_copy(t, x.data[:, off:off + w])
        outs.append(Var(t))
        off += w

    def bwd() -> None:
        if not x.needs_grad or all(o.grad is None for o in outs):
            return
        g = K.zeros((m, sum(widths)), x.data.dtype, x.data.device)
        o0 = 0
        for o, w in zip(outs, widths):
            if o.grad is not None:
                src = o.grad if o.grad.dtype == g.dtype else K.cast(o.grad.contiguous(), g.dtype)
                K.row_copy(g[:, o0:o0 + w], src.contiguous())
            o0 += w
        x.add_grad(g)

    tape.record(bwd)
    return outs


def op_attention(tape: Tape, q: Var, k: Var, v: Var, batch: int, heads: int) -> Var:
    """softmax(q k^T / sqrt(head_dim)) v on GIVEN projections (``cinema/vit.py:505-517``): q bf16 [b*tq, c], k / v bf16 [b*tk, c] -> bf16 [b*tq, c].  The fused
    ops of ``ops_block`` project inside; this one is for q / k that went through ``q_norm`` / ``k_norm`` first."""
    c = q.data.shape[1]
    tq, tk = q.data.shape[0] // batch, k.data.shape[0] // batch
    scale = (c // heads) ** -0.5
    q3, k3, v3 = q.data.view(batch, tq, c), k.data.view(batch, tk, c), v.data.view(batch, tk, c)
    if tape.train and T.ATTN_O_LO:
        o, lse, o_lo = K.attention_fwd(q3, k3, v3, heads, scale, want_lo=True)
    else:
        (o, lse), o_lo = K.attention_fwd(q3, k3, v3, heads, scale), None
    y = Var(o.view(batch * tq, c))

    def bwd() -> None:
        if y.grad is None:
            return
        dq, dk, dv = (K.empty((batch, t, c), dtype=BF16, device=o.device) for t in (tq, tk, tk))
        K.attention_bwd(q3, k3, v3, o, y.grad_bf16().contiguous().view(batch, tq, c), lse, heads, scale, dq, dk, dv, o_lo=o_lo)
        q.add_grad(dq.reshape(batch * tq, c))
        k.add_grad(dk.reshape(batch * tk, c))
        v.add_grad(dv.reshape(batch * tk, c))

    tape.record(bwd)
    return y
