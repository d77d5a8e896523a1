"""Tape ops of the convolutional parts: dense and visible-voxel depthwise convolutions, the fused MaskedConvBlock of the stems, implicit-GEMM 'same' convolutions, transposed convolutions.

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
    BF16, F32, Tape, Var, WEIGHTS, _split_k_conv, _wgrad_launch, const, conv_same_grad_to_param, w_conv_same, w_plain, wgrad,
)

__all__ = ['_chan_last_strides', '_op_conv1ch', 'conv_transpose_grad_to_param', 'conv_zblock', 'op_conv_same', 'op_conv_transpose', 'op_dwconv', 'op_sparse_dwconv', 'op_stem_block', 'stem_block_ok', 'w_conv_transpose']


def op_dwconv(tape: Tape, x: Var, spatial: tuple, weight: torch.nn.Parameter, bias: torch.nn.Parameter | None,
              in_mask: torch.Tensor | None = None) -> Var:
    """Depthwise 5^n conv on bf16 channels-last rows [b*prod(spatial), c].  ``in_mask`` (uint8 per voxel) marks voxels whose
    *input* was zeroed by the caller's mask multiply; the data gradient is zeroed there too (conv.py:410-411)."""
    c = x.data.shape[1]
    b = x.data.shape[0] // int(torch.Size(spatial).numel())
    xs = x.data.view(b, *spatial, c)
    y = Var(K.dwconv_fwd(xs, weight.detach(), None if bias is None else bias.detach()).view(-1, c))
    wv, bv = tape.pvar(weight), tape.pvar(bias)

    def bwd() -> None:
        if y.grad is None:
            return
        dy = y.grad.view(b, *spatial, c)
        K.dwconv_bwd_weight(xs, dy, wv.grad_buffer(tuple(weight.shape)), None if bias is None else bv.grad_buffer((c,)))
        if x.needs_grad:
            x.add_grad(K.dwconv_bwd_data(dy, weight.detach(), in_mask).view(-1, c))

    tape.record(bwd)
    return y


def op_sparse_dwconv(tape: Tape, x: Var, geom, weight: torch.nn.Parameter, bias: torch.nn.Parameter | None) -> Var:  # noqa: ANN001
    """Depthwise conv on the visible-voxel compact rows of an MAE step (``hip.sparse_geom``): exactly the dense masked conv
    of ``op_dwconv`` evaluated at the visible voxels, whose inputs at masked voxels are zero by construction."""
    y = Var(K.sparse_dwconv(x.data, weight.detach(), None if bias is None else bias.detach(), geom))
    wv, bv = tape.pvar(weight), tape.pvar(bias)

    def bwd() -> None:
        if y.grad is None:
            return
        c = x.data.shape[1]
        dw, db, dy = wv.grad_buffer(tuple(weight.shape)), None if bias is None else bv.grad_buffer((c,)), y.grad
        _wgrad_launch(lambda: K.sparse_dwconv_bwd_weight(x.data, dy, tuple(weight.shape), dw, db, geom), x.data, dy, alt=1,
                      keys=(dw.data_ptr(),) if db is None else (dw.data_ptr(), db.data_ptr()))  # off the critical path
        if x.needs_grad:
            x.add_grad(K.sparse_dwconv(y.grad, weight.detach(), None, geom, flip=True))

    tape.record(bwd)
    return y


def stem_block_ok(x: Var, c: int, hidden: int) -> bool:
    return T.FUSED_STEM and x.data.is_cuda and x.data.dtype == F32 and x.data.is_contiguous() and K.stem_supported(c, hidden) and not K.FORCE_GENERIC


def op_stem_block(tape: Tape, x: Var, geom, blk) -> Var:  # noqa: ANN001
    """One MaskedConvBlock (``cinema/conv.py:405-413``) on visible-voxel compact rows x fp32 [rows, c] as three launches: LN1 -> conv1 (``stem_ln_linear``), the
    depthwise 5^n conv over the neighbour lists, conv2 + residual -> LN2 -> fc1 -> GELU -> fc2 + residual (``stem_mlp_fwd``: the 4c-wide hidden layer stays in
    registers).  Backward: ``stem_mlp_bwd`` (recomputes the hidden layer from the saved x1), the depthwise data / weight gradients, ``stem_ln_linear_bwd``, and the
    four 1x1-convolution weight gradients + biases as ONE ``stem_wgrad`` launch on the weight-gradient stream."""
    n1, n2, c1, c2, dw, fc1, fc2 = blk.norm1, blk.norm2, blk.conv1, blk.conv2, blk.dw_conv, blk.mlp.fc1, blk.mlp.fc2
    w1, w2, wf1, wf2 = w_plain(c1.weight), w_plain(c2.weight), w_plain(fc1.weight), w_plain(fc2.weight)
    det = lambda t: t.detach()  # noqa: E731
    xn, h = K.stem_ln_linear(x.data, det(n1.weight), det(n1.bias), n1.eps, w1, det(c1.bias), want_xn=tape.train)
    d = K.sparse_dwconv(h, det(dw.weight), None if dw.bias is None else det(dw.bias), geom)
    x1, x2 = K.stem_mlp_fwd(d, x.data, w2, det(c2.bias), det(n2.weight), det(n2.bias), n2.eps, wf1, det(fc1.bias), wf2, det(fc2.bias), want_x1=tape.train)
    y = Var(x2)
    pv = {p_: tape.pvar(p_) for p_ in (n1.weight, n1.bias, n2.weight, n2.bias, c1.weight, c1.bias, c2.weight, c2.bias, dw.weight, dw.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias)
          if p_ is not None}
    c = x.data.shape[1]

    def gbuf(p_: torch.nn.Parameter | None, shape: tuple) -> torch.Tensor | None:
        return pv[p_].grad_buffer(shape) if (p_ is not None and p_.requires_grad) else None

    def bwd() -> None:
        if y.grad is None:
            return
        g2 = y.grad if y.grad.dtype == F32 else K.cast(y.grad, F32)
        o = K.stem_mlp_bwd(g2.contiguous(), x1, w2, det(n2.weight), det(n2.bias), n2.eps, wf1, det(fc1.bias), wf2)
        part2, n_part2 = o["partials"]
        if n2.weight.requires_grad or n2.bias.requires_grad:
            tape.pending_ln.append((part2, n_part2, c, gbuf(n2.weight, (c,)), gbuf(n2.bias, (c,))))
        dd = o["dd"]
        if dw.weight.requires_grad:
            dwg, dbg = gbuf(dw.weight, tuple(dw.weight.shape)), gbuf(dw.bias, (c,))
            _wgrad_launch(lambda: K.sparse_dwconv_bwd_weight(h, dd, tuple(dw.weight.shape), dwg, dbg, geom), h, dd, alt=1,
                          keys=(dwg.data_ptr(),) if dbg is None else (dwg.data_ptr(), dbg.data_ptr()))
        dh = K.sparse_dwconv(dd, det(dw.weight), None, geom, flip=True)
        dx, (part1, n_part1) = K.stem_ln_linear_bwd(dh, x.data, o["dx1"], det(n1.weight), n1.eps, w1)
        if n1.weight.requires_grad or n1.bias.requires_grad:
            tape.pending_ln.append((part1, n_part1, c, gbuf(n1.weight, (c,)), gbuf(n1.bias, (c,))))
        probs = [(o["g2_16"], o["a"], fc2, wf2), (o["dz"], o["xn2"], fc1, wf1), (o["dx1_16"], d, c2, w2), (dh, xn, c1, w1)]
        probs = [(dy, xx, gbuf(m.weight, tuple(w16.shape)), gbuf(m.bias, (w16.shape[0],))) for dy, xx, m, w16 in probs if m.weight.requires_grad]
        if probs:
            _wgrad_launch(lambda: K.stem_wgrad(probs), *[t for pr in probs for t in pr[:2]], alt=1, keys=tuple(t.data_ptr() for pr in probs for t in pr[2:] if t is not None))
        if x.needs_grad:
            x.add_grad(dx)

    tape.record(bwd)
    return y


def op_conv_same(tape: Tape, x: Var, batch: int, spatial: tuple, weight: torch.nn.Parameter, bias: torch.nn.Parameter | None, *,
                 residual: Var | None = None, out_f32: bool = False) -> Var:
    """Dense "same"-padded conv on bf16 channels-last rows x [batch*prod(spatial), c] (``ConvResBlock`` convs, ``cinema/conv.py:320-345``):
    im2col + MFMA GEMM (+ bias, + fp32 residual).  The column matrix is not kept: the backward pass rebuilds it for the weight gradient."""
    ks = tuple(weight.shape[2:])
    c = x.data.shape[1]
    c_out = weight.shape[0]
    if (c == 1 and weight.shape[1] == 1 and out_f32 and residual is None and x.data.is_cuda and x.data.dtype == BF16 and x.data.is_contiguous()
            and all(k in (1, 3) for k in ks) and K.fanout_ok(c_out, 1)):
        return _op_conv1ch(tape, x, batch, spatial, weight, bias)  # one input channel (the raw-image block): direct stencil kernels, no im2col
    w16 = w_conv_same(weight)
    xs = x.data.view(batch, *spatial, c)
    dev = x.data.device
    # implicit GEMM (cinema_conv_gemm_bf16): the MFMA kernel gathers its A tiles from the volume, the 27x im2col matrix is never written; the
    # 1-channel raw-image block (c = 1) and exotic kernel extents keep the im2col path
    implicit = x.data.is_cuda and c % 8 == 0 and c_out % 8 == 0 and all(k in (1, 3) for k in ks) and (residual is None or residual.data.dtype == F32)
    # narrow layers (c_out <= 64): z-blocked form - a GEMM row is a group of 2 / 4 consecutive z voxels and n = zb * c_out fills the 128-wide tile
    # (c_out = 32: 50 % useful MACs instead of 25 %, half the gathered bytes per output); see cinema_conv_gemm_bf16
    zb_f = conv_zblock(c_out, ks, spatial) if implicit else 1
    zb_d = conv_zblock(c, ks, spatial) if implicit else 1
    taps = None
    if implicit and zb_f > 1:
        pkey = (weight,) if bias is None else (weight, bias)
        wz, bz = WEIGHTS.get(pkey, f"conv_zb{zb_f}", lambda: K.conv_weight_zblock(w16, c, zb_f, False, None if bias is None else bias.detach()))
        taps_z = const(("conv_taps", c, ks, tuple(spatial), wz.shape[1], False, str(dev), zb_f), lambda: K.conv_tap_table(c, ks, spatial, wz.shape[1], False, dev, zb=zb_f))
        y = Var(K.conv_gemm(xs, wz, taps_z, bias=bz, residual=None if residual is None else residual.data.contiguous(),
                            out_dtype=F32 if (out_f32 or residual is not None) else BF16, zb=zb_f).view(-1, c_out))
    elif implicit:
        taps = const(("conv_taps", c, ks, tuple(spatial), w16.shape[1], False, str(dev)), lambda: K.conv_tap_table(c, ks, spatial, w16.shape[1], False, dev))
        y = Var(K.conv_gemm(xs, w16, taps, bias=None if bias is None else bias.detach(), residual=None if residual is None else residual.data,
                            out_dtype=F32 if (out_f32 or residual is not None) else BF16))
    else:
        cols = K.im2col(xs, ks)
        y = Var(K.gemm(cols, w16, bias=None if bias is None else bias.detach(), residual=None if residual is None else residual.data,
                       out_dtype=F32 if (out_f32 or residual is not None) else BF16))
        del cols
    wv, bv = tape.pvar(weight), tape.pvar(bias)

    def bwd() -> None:
        if y.grad is None:
            return
        if residual is not None:
            residual.add_grad(y.grad, y.grad16)
        dy16 = y.grad_bf16()
        if weight.requires_grad and implicit and zb_f > 1:  # z-blocked: R [zb c_out, 9 (zb + 2) c] = dy_groups^T im2col_zb(x), its zb bands folded into dW
            coords = const(("conv_coords", batch, tuple(spatial), str(dev), zb_f), lambda: K.conv_coord_table(batch, spatial, dev, zb=zb_f))
            dst = wv.grad_buffer(tuple(w16.shape), conv_same_grad_to_param(weight))
            db = bv.grad_buffer((c_out,)) if (bias is not None and bias.requires_grad) else None
            dyz = dy16.contiguous().view(-1, zb_f * c_out)
            ldz = 9 * (zb_f + 2) * c
            split = _split_k_conv(dyz.shape[0], zb_f * c_out, ldz)

            # scratch of the side-stream launches: allocated here and handed over as operands, so that it lives until the side stream has been joined
            r = K.empty((zb_f * c_out, ldz), dtype=F32, device=dev)
            rs = K.zeros(zb_f * c_out, device=dev) if db is not None else None

            def launch() -> None:
                K.conv_wgrad(dyz, xs, taps_z, coords, r, split, a_rowsum=rs, zb=zb_f, accumulate=False)
                K.conv_wgrad_zfold(r, c_out, c, zb_f, dst, rs, db)

            _wgrad_launch(launch, dyz, xs, *((r,) if rs is None else (r, rs)), keys=tuple(t.data_ptr() for t in (dst, db) if t is not None))
        elif weight.requires_grad and implicit:  # dW = dy^T im2col(x) with the column matrix gathered inside the GEMM (cinema_conv_wgrad_bf16)
            coords = const(("conv_coords", batch, tuple(spatial), str(dev)), lambda: K.conv_coord_table(batch, spatial, dev))
            dst = wv.grad_buffer(tuple(w16.shape), conv_same_grad_to_param(weight))
            db = bv.grad_buffer((c_out,)) if (bias is not None and bias.requires_grad) else None
            dyc = dy16.contiguous()
            split = _split_k_conv(dyc.shape[0], c_out, w16.shape[1])
            _wgrad_launch(lambda: K.conv_wgrad(dyc, xs, taps, coords, dst, split, a_rowsum=db), dyc, xs, keys=tuple(t.data_ptr() for t in (dst, db) if t is not None))
        elif weight.requires_grad:
            wgrad(tape, dy16, K.im2col(xs, ks), wv, bv if (bias is not None and bias.requires_grad) else None, tuple(w16.shape),
                  conv_same_grad_to_param(weight))
        if x.needs_grad:
            if implicit:  # data gradient = the same implicit convolution on dy with transposed weights and negated tap offsets (no col2im pass)
                wt = WEIGHTS.get((weight,), "conv_dgrad", lambda: K.conv_weight_dgrad(weight.detach()))
                if zb_d > 1:
                    wtz = WEIGHTS.get((weight,), f"conv_dgrad_zb{zb_d}", lambda: K.conv_weight_zblock(wt, c_out, zb_d, True)[0])
                    taps_t = const(("conv_taps", c_out, ks, tuple(spatial), wtz.shape[1], True, str(dev), zb_d),
                                   lambda: K.conv_tap_table(c_out, ks, spatial, wtz.shape[1], True, dev, zb=zb_d))
                    x.add_grad(K.conv_gemm(dy16.contiguous().view(batch, *spatial, c_out), wtz, taps_t, zb=zb_d).view(-1, c))
                else:
                    taps_t = const(("conv_taps", c_out, ks, tuple(spatial), wt.shape[1], True, str(dev)),
                                   lambda: K.conv_tap_table(c_out, ks, spatial, wt.shape[1], True, dev))
                    x.add_grad(K.conv_gemm(dy16.contiguous().view(batch, *spatial, c_out), wt, taps_t))
            else:
                dcols = K.gemm(dy16, w16, a_kmajor=True, b_kmajor=False)
                x.add_grad(K.col2im(dcols, (batch, *spatial, c), ks).view(-1, c))

    tape.record(bwd)
    return y


def _op_conv1ch(tape: Tape, x: Var, batch: int, spatial: tuple, weight: torch.nn.Parameter, bias: torch.nn.Parameter | None) -> Var:
    """"Same" conv of a one-channel volume on the fp32 master weight (``cinema_conv1ch_fwd / bwd``): fp32 rows [batch * prod(spatial), c_out]."""
    xs = x.data.view(batch, *spatial)
    y = Var(K.conv1ch_fwd(xs, weight.detach(), None if bias is None else bias.detach()))
    wv, bv = tape.pvar(weight), tape.pvar(bias)

    def bwd() -> None:
        if y.grad is None:
            return
        dw = wv.grad_buffer(tuple(weight.shape)) if weight.requires_grad else None
        db = bv.grad_buffer((weight.shape[0],)) if (bias is not None and bias.requires_grad) else None
        g = y.grad if y.grad.dtype == F32 else K.cast(y.grad, F32)
        if dw is not None or db is not None:
            _wgrad_launch(lambda: K.conv1ch_bwd(xs, weight.detach(), g.contiguous(), dw, db, want_dx=False), xs, g, keys=tuple(t.data_ptr() for t in (dw, db) if t is not None))
        if x.needs_grad:  # data gradient: dy @ w as one skinny MFMA GEMM + the mirrored gather (measured: 0.30 ms against 0.56-1.1 ms for a direct stencil)
            w16 = w_conv_same(weight)
            dcols = K.gemm(y.grad_bf16(), w16, a_kmajor=True, b_kmajor=False)
            x.add_grad(K.col2im(dcols, (batch, *spatial, 1), tuple(weight.shape[2:])).view(-1, 1))

    tape.record(bwd)
    return y




def conv_zblock(n_out: int, ks: tuple, spatial: tuple) -> int:
    """z-blocking factor of the implicit convolution for a layer with ``n_out`` output channels: the largest of 4 / 2 that keeps zb * n_out within the
    128-wide tile and divides Z (3x3x3 kernels on 3-D volumes only); 1 = plain."""
    if len(spatial) != 3 or tuple(ks) != (3, 3, 3):
        return 1
    for zb in (4, 2):
        if zb * n_out <= 128 and spatial[2] % zb == 0:
            return zb
    return 1


def _chan_last_strides(chans: int, spatial: tuple) -> tuple:
    sp, acc = [], chans
    for d in reversed(spatial):
        sp.append(acc)
        acc *= d
    return (acc, 1, *reversed(sp))


def w_conv_transpose(weight: torch.nn.Parameter) -> torch.Tensor:
    """ConvTranspose weight (c_in, c_out, *k) -> bf16 [(*k, c_out), c_in]: one GEMM gives every output voxel of a k == s up-sampling."""

    return WEIGHTS.get((weight,), "conv_transpose", lambda: K.convt_weight_rows(weight.detach())[0])


def conv_transpose_grad_to_param(weight: torch.nn.Parameter) -> Callable:
    shape = weight.shape

    def conv(g: torch.Tensor) -> torch.Tensor:
        nd = len(shape) - 2
        g = g.reshape(*shape[2:], shape[1], shape[0])
        return g.permute(nd + 1, nd, *range(nd)).contiguous()

    conv.hip_relayout = ("convt",)  # the flat-buffer path adds the rows with one re-layout kernel (cinema_convt_weight_relayout) instead of this permute
    return conv


def op_conv_transpose(tape: Tape, x: Var, batch: int, spatial: tuple, weight: torch.nn.Parameter, bias: torch.nn.Parameter | None,
                      skip: Var | None = None) -> tuple:
    """k == s transposed conv (``UpsampleDecoder.up``, ``cinema/segmentation/convunetr.py:62-101``) on bf16 channels-last rows
    x [batch*prod(spatial), c_in]: one GEMM to rows [(*k, c_out)] per input voxel, scattered to the up-sampled fp32 volume
    (+ ``skip``, the encoder feature added right after the up-sampling).  Returns (Var fp32 [batch*prod(out_spatial), c_out], out_spatial)."""
    ks = tuple(weight.shape[2:])
    c_out = weight.shape[1]
    k_vol = math.prod(ks)
    wt = w_conv_transpose(weight)
    bias_t = None if bias is None else WEIGHTS.get((weight, bias), f"tile{k_vol}", lambda: K.convt_weight_rows(weight.detach(), bias.detach())[1])
    rows = K.gemm(x.data, wt, bias=bias_t)
    out_spatial = tuple(s * k for s, k in zip(spatial, ks))
    n_out = batch * math.prod(out_spatial)
    geom = K.patch_geom(batch, c_out, spatial, ks, _chan_last_strides(c_out, out_spatial))
    if skip is not None:
        dst = K.empty((n_out, c_out), dtype=F32, device=x.data.device)
        K.row_copy(dst, skip.data)
        K.patch_scatter(rows, dst, geom, accumulate=True)
    else:
        dst = K.empty((n_out, c_out), dtype=F32, device=x.data.device)
        K.patch_scatter(rows, dst, geom)
    y = Var(dst)
    wv, bv = tape.pvar(weight), tape.pvar(bias)

    def bwd() -> None:
        if y.grad is None:
            return
        if skip is not None:
            skip.add_grad(y.grad, y.grad16)
        drows = K.patch_gather(y.grad, geom, BF16)  # [n_in_vox, k_vol * c_out]
        if weight.requires_grad:
            wgrad(tape, drows, x.data, wv, None, tuple(wt.shape), conv_transpose_grad_to_param(weight))
            if bias is not None and bias.requires_grad:
                K.colsum(drows.view(-1, c_out), bv.grad_buffer((c_out,)))
        if x.needs_grad:
            x.add_grad(K.gemm(drows, wt, a_kmajor=True, b_kmajor=False))

    tape.record(bwd)
    return y, out_spatial
