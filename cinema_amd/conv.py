"""Convolutional building blocks of the conv stem (interface of the reference ``cinema/conv.py``).

The classes keep the reference's constructor signatures, attribute names and parameter shapes (so that ``state_dict``
keys, seeded initialisation and ``set_grad_ckpt`` calls are interchangeable), but their compute is a sequence of HIP
kernels on channels-last rows recorded on a :class:`cinema_amd.tape.Tape`:

* ``ConvNormActBlock`` with ``kernel == stride`` (``cinema/conv.py:212-273`` as used at ``cinema/convvit.py:94-102``):
  patch gather -> MFMA GEMM (+bias) -> fused LayerNorm+GELU.
* ``MaskedConvBlock`` (``cinema/conv.py:349-415``): LN -> 1x1 GEMM (+bias, visible-mask rows) -> depthwise 5^n conv ->
  1x1 GEMM (+bias, +residual) ; LN -> fc1 GEMM (+bias, GELU) -> fc2 GEMM (+bias, +residual).

Activation checkpointing (``grad_ckpt``) is accepted and ignored: 288 GB of HBM3E holds every activation of the
largest BASELINE config, so nothing is recomputed.
"""

from __future__ import annotations

import math

import torch
from torch import nn

from cinema_amd import hip as K
from cinema_amd import tape as T

KernelSizeType = tuple | int


class _CkptFlag:
    """``set_grad_ckpt`` API of the reference (``cinema/conv.py:29-32``); a recorded no-op here."""

    grad_ckpt = False

    def set_grad_ckpt(self, enable: bool = True) -> None:
        self.grad_ckpt = enable


def _standalone_linear(mod: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """y = x W^T + b on the HIP GEMM for a direct module call (fp32 in/out, bf16 MFMA compute)."""
    shape = x.shape

    def run(tp: T.Tape, xv: T.Var):  # noqa: ANN202
        return [T.op_linear(tp, T.op_cast_bf16(tp, xv), mod.weight, mod.bias, out_f32=True)], []

    (y,) = T.taped_call(run, [x.float().reshape(-1, shape[-1]).contiguous()], [mod.weight, mod.bias])
    return y.reshape(*shape[:-1], -1)


class Linear(nn.Linear, _CkptFlag):
    """``nn.Linear`` parameter container whose forward is the bf16 MFMA GEMM (reference ``cinema/conv.py:21-36``)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _standalone_linear(self, x)


class _ConvBase(_CkptFlag):
    def forward(self, x: torch.Tensor) -> torch.Tensor:  # noqa: ARG002
        raise NotImplementedError(
            f"{type(self).__name__} is a parameter container in cinema_amd: its compute is fused into the owning block's HIP "
            "kernel sequence (call the enclosing ConvNormActBlock / MaskedConvBlock / DownsampleEncoder instead)."
        )


class Conv2d(_ConvBase, nn.Conv2d):
    """Parameter container (reference ``cinema/conv.py:39-54``)."""


class Conv3d(_ConvBase, nn.Conv3d):
    """Parameter container (reference ``cinema/conv.py:57-72``)."""


class ConvTranspose2d(_ConvBase, nn.ConvTranspose2d):
    """Parameter container (reference ``cinema/conv.py:75-90``)."""


class ConvTranspose3d(_ConvBase, nn.ConvTranspose3d):
    """Parameter container (reference ``cinema/conv.py:93-108``)."""


class ConvLayerNorm(nn.LayerNorm):
    """LayerNorm over the channel axis of a channels-first tensor (reference ``cinema/conv.py:169-187``)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        c = x.shape[1]
        rows = x.movedim(1, -1).contiguous()
        shape = rows.shape

        def run(tp: T.Tape, xv: T.Var):  # noqa: ANN202
            return [T.op_layernorm(tp, xv, self.weight, self.bias, self.eps, out_f32=True)], []

        (y,) = T.taped_call(run, [rows.float().reshape(-1, c)], [self.weight, self.bias])
        return y.reshape(shape).movedim(-1, 1).contiguous()


def get_conv_norm(n_dims: int, in_chans: int, norm: str, eps: float = 1e-6, n_groups: int = 32) -> nn.Module:  # noqa: ARG001
    """Reference ``cinema/conv.py:190-209``; only ``'layer'`` (used by every released model) has a HIP path."""
    if norm == "layer":
        return ConvLayerNorm(in_chans, eps=eps)
    if norm in ("instance", "group"):
        raise NotImplementedError(f"norm='{norm}' is not on the MI355X hot path (all CineMA models use 'layer').")
    raise ValueError(f"Invalid norm type, got {norm}, must be 'instance' or 'layer' or 'group'.")


class Volume:
    """A channels-last activation on the tape: ``var.data`` is [batch * prod(spatial), chans] (fp32 residual stream)."""

    def __init__(self, var: T.Var, batch: int, spatial: tuple, chans: int) -> None:
        self.var, self.batch, self.spatial, self.chans = var, batch, tuple(spatial), chans

    def strides(self) -> tuple:
        """Element strides (batch, channel, *spatial) of the channels-last layout."""
        sp = []
        acc = self.chans
        for s in reversed(self.spatial):
            sp.append(acc)
            acc *= s
        return (acc, 1, *reversed(sp))


class CompactVolume:
    """Visible-voxel activation of the conv stem in an MAE step: ``var.data`` is [n_tok * prod(block), chans] (fp32), token-major
    compact rows (see ``csrc/sparse_conv.hip``); ``geom`` is the ``hip.sparse_geom`` of this stage."""

    def __init__(self, var: T.Var, n_tok: int, block: tuple, chans: int, geom, pos: torch.Tensor, inv_pos: torch.Tensor) -> None:  # noqa: ANN001
        self.var, self.n_tok, self.block, self.chans, self.geom, self.pos, self.inv_pos = var, n_tok, tuple(block), chans, geom, pos, inv_pos

    @property
    def block_voxels(self) -> int:
        return math.prod(self.block)

    def token_rows(self, tp: T.Tape) -> T.Var:
        """bf16 [n_tok, block_voxels * chans]: one row per kept token (GEMM operand of a k == s conv over the token's block)."""
        return T.op_cast_bf16(tp, T.op_view(tp, self.var, (self.n_tok, self.block_voxels * self.chans)))


class ConvNormActBlock(nn.Module, _CkptFlag):
    """conv(kernel == stride, 'valid') -> ConvLayerNorm -> GELU  (reference ``cinema/conv.py:212-273``)."""

    def __init__(self, n_dims: int, in_chans: int, out_chans: int, norm: str, kernel_size: KernelSizeType = 3, stride: KernelSizeType = 1,
                 padding: str = "same", act_layer: type = nn.GELU) -> None:
        if n_dims not in {2, 3}:
            raise ValueError(f"Invalid n_dims, must be 2 or 3, got {n_dims}.")
        if not isinstance(kernel_size, int) and len(kernel_size) != n_dims:
            raise ValueError(f"Invalid kernel_size {kernel_size}, must be an integer or a tuple of {n_dims} integers.")
        if not isinstance(stride, int) and len(stride) != n_dims:
            raise ValueError(f"Invalid stride {stride}, must be an integer or a tuple of {n_dims} integers.")
        super().__init__()
        conv_cls = Conv2d if n_dims == 2 else Conv3d
        self.conv = conv_cls(in_chans, out_chans, kernel_size=kernel_size, stride=stride, padding=padding)
        self.norm = get_conv_norm(n_dims=n_dims, in_chans=out_chans, norm=norm)
        self.act = act_layer()
        if tuple(self.conv.kernel_size) != tuple(self.conv.stride) or padding != "valid" or not isinstance(self.act, nn.GELU):
            raise NotImplementedError("cinema_amd implements the non-overlapping (kernel == stride, 'valid', GELU) form used by the conv stem.")

    def set_grad_ckpt(self, enable: bool = True) -> None:
        self.grad_ckpt = enable
        self.conv.set_grad_ckpt(enable)

    def tape_forward(self, tp: T.Tape, src: T.Var, batch: int, chans: int, spatial: tuple, strides: tuple) -> Volume:
        """``src`` holds a (b, chans, *spatial) volume addressed by ``strides`` (channels-first image or channels-last rows)."""
        patch = tuple(self.conv.kernel_size)
        for s, p in zip(spatial, patch):
            if s % p != 0:
                raise ValueError(f"Input size ({spatial}) should be divisible by patch size ({patch}).")
        grid = tuple(s // p for s, p in zip(spatial, patch))
        geom = K.patch_geom(batch, chans, grid, patch, strides)
        rows = T.op_patch_gather(tp, src, geom)
        y = T.op_linear(tp, rows, self.conv.weight, self.conv.bias, w16=T.w_patch(self.conv.weight),
                        to_param_layout=T.patch_grad_to_param(self.conv.weight))
        out = T.op_layernorm(tp, y, self.norm.weight, self.norm.bias, self.norm.eps, act=1, out_f32=True)
        return Volume(out, batch, grid, self.conv.out_channels)

    def tape_forward_rows(self, tp: T.Tape, rows: T.Var) -> T.Var:
        """The same block on already gathered bf16 patch rows [n, prod(kernel) * in_chans] (visible-voxel stem)."""
        y = T.op_linear(tp, rows, self.conv.weight, self.conv.bias, w16=T.w_patch(self.conv.weight),
                        to_param_layout=T.patch_grad_to_param(self.conv.weight))
        return T.op_layernorm(tp, y, self.norm.weight, self.norm.bias, self.norm.eps, act=1, out_f32=True)


class ConvMlp(nn.Module, _CkptFlag):
    """1x1-conv MLP, parameter container (reference ``cinema/conv.py:111-166``; timm ``Mlp(use_conv=True)`` layout)."""

    def __init__(self, n_dims: int, in_features: int, hidden_features: int | None = None, out_features: int | None = None,
                 act_layer: type = nn.GELU, norm_layer: type | None = None, bias: tuple | bool = True, drop: tuple | float = 0.0) -> None:
        if n_dims not in {2, 3}:
            raise ValueError(f"Invalid n_dims, must be 2 or 3, got {n_dims}.")
        super().__init__()
        hidden_features = hidden_features or in_features
        out_features = out_features or in_features
        bias_t = tuple(bias) if isinstance(bias, (tuple, list)) else (bias, bias)
        drop_t = tuple(drop) if isinstance(drop, (tuple, list)) else (drop, drop)
        if norm_layer is not None or any(d > 0 for d in drop_t) or act_layer is not nn.GELU:
            raise NotImplementedError("cinema_amd ConvMlp: GELU, no inner norm, no dropout (what every CineMA model uses).")
        # the reference first builds timm's 2-D 1x1 convs and then replaces them (conv.py:146-159); draw the same
        # random numbers so that seeded construction stays identical to the reference
        nn.Conv2d(in_features, hidden_features, kernel_size=1, bias=bias_t[0])
        nn.Conv2d(hidden_features, out_features, kernel_size=1, bias=bias_t[1])
        conv_cls = Conv2d if n_dims == 2 else Conv3d
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop_t[0])
        self.norm = nn.Identity()
        self.drop2 = nn.Dropout(drop_t[1])
        self.fc1 = conv_cls(in_features, hidden_features, kernel_size=1, bias=bias_t[0])
        self.fc2 = conv_cls(hidden_features, out_features, kernel_size=1, bias=bias_t[1])

    def set_grad_ckpt(self, enable: bool = True) -> None:
        self.grad_ckpt = enable
        self.fc1.set_grad_ckpt(enable)
        self.fc2.set_grad_ckpt(enable)


class MaskedConvBlock(nn.Module, _CkptFlag):
    """ConvMAE block (reference ``cinema/conv.py:349-415``)."""

    def __init__(self, n_dims: int, in_chans: int, mlp_ratio: int = 4, dropout: float = 0.0, drop_path: float = 0.0, act_layer: type = nn.GELU,
                 norm: str = "layer") -> None:
        if n_dims not in {2, 3}:
            raise ValueError(f"Invalid n_dims, must be 2 or 3, got {n_dims}.")
        super().__init__()
        if drop_path > 0.0:
            raise NotImplementedError("drop_path > 0 is a fine-tuning option outside the MAE pre-training path.")
        self.norm1 = get_conv_norm(n_dims=n_dims, in_chans=in_chans, norm=norm)
        self.norm2 = get_conv_norm(n_dims=n_dims, in_chans=in_chans, norm=norm)
        conv_cls = Conv2d if n_dims == 2 else Conv3d
        self.conv1 = conv_cls(in_chans, in_chans, kernel_size=1, padding="same")
        self.conv2 = conv_cls(in_chans, in_chans, kernel_size=1, padding="same")
        self.dw_conv = conv_cls(in_chans, in_chans, kernel_size=5, padding="same", groups=in_chans)
        self.drop_path = nn.Identity()
        self.mlp = ConvMlp(n_dims=n_dims, in_features=in_chans, hidden_features=in_chans * mlp_ratio, act_layer=act_layer, drop=dropout)

    def set_grad_ckpt(self, enable: bool = True) -> None:
        self.grad_ckpt = enable
        for m in (self.conv1, self.conv2, self.dw_conv, self.mlp):
            m.set_grad_ckpt(enable)

    def tape_forward(self, tp: T.Tape, x: Volume, vis: torch.Tensor | None) -> Volume:
        """``vis``: uint8 [batch * prod(spatial)], 1 = visible (the reference's ``mask`` argument, ``conv.py:405``)."""
        xn = T.op_layernorm(tp, x.var, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        h = T.op_linear(tp, xn, self.conv1.weight, self.conv1.bias, row_mask=vis)
        h = T.op_dwconv(tp, h, x.spatial, self.dw_conv.weight, self.dw_conv.bias, in_mask=vis)
        x1 = T.op_linear(tp, h, self.conv2.weight, self.conv2.bias, residual=x.var)
        xn2 = T.op_layernorm(tp, x1, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        x2 = T.op_mlp(tp, xn2, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias, residual=x1)
        return Volume(x2, x.batch, x.spatial, x.chans)

    def tape_forward_compact(self, tp: T.Tape, x: CompactVolume) -> CompactVolume:
        """The block on the visible voxels only: every row is visible, so the mask multiply disappears and the depthwise conv
        looks its neighbours up through the token rank map (masked neighbours contribute the zeros the reference multiplies in)."""
        if T.stem_block_ok(x.var, x.chans, self.mlp.fc1.weight.shape[0]) and self.conv1.bias is not None and self.mlp.fc1.bias is not None:
            # c = 64 / 128 (every released CineMA stem): the per-voxel halves of the block as fused kernels (csrc/stem.hip)
            return CompactVolume(T.op_stem_block(tp, x.var, x.geom, self), x.n_tok, x.block, x.chans, x.geom, x.pos, x.inv_pos)
        xn = T.op_layernorm(tp, x.var, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        h = T.op_linear(tp, xn, self.conv1.weight, self.conv1.bias)
        h = T.op_sparse_dwconv(tp, h, x.geom, self.dw_conv.weight, self.dw_conv.bias)
        x1 = T.op_linear(tp, h, self.conv2.weight, self.conv2.bias, residual=x.var)
        xn2 = T.op_layernorm(tp, x1, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        x2 = T.op_mlp(tp, xn2, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias, residual=x1)
        return CompactVolume(x2, x.n_tok, x.block, x.chans, x.geom, x.pos, x.inv_pos)

    def forward(self, x: torch.Tensor, mask: torch.Tensor | None = None) -> torch.Tensor:
        """Channels-first (b, C, *S) in/out like the reference; layout converted at the boundary."""
        b, c, *sp = x.shape
        rows = x.movedim(1, -1).contiguous().float().reshape(-1, c)
        vis = None if mask is None else mask.reshape(-1).to(torch.uint8).contiguous()

        def run(tp: T.Tape, xv: T.Var):  # noqa: ANN202
            return [self.tape_forward(tp, Volume(xv, b, tuple(sp), c), vis).var], []

        (y,) = T.taped_call(run, [rows], list(self.parameters()))
        return y.reshape(b, *sp, c).movedim(-1, 1).contiguous()


def n_voxels(spatial: tuple) -> int:
    return math.prod(spatial)


class ConvResBlock(nn.Module, _CkptFlag):
    """norm -> GELU -> conv -> norm -> GELU -> (dropout) -> conv, plus a 1x1 shortcut when the channel count changes
    (reference ``cinema/conv.py:276-348``).  The two dense "same" convs run as im2col + MFMA GEMM (``tape.op_conv_same``), the
    LayerNorm + GELU pairs are one fused kernel each, the shortcut is the fp32 residual of the second GEMM."""

    def __init__(self, n_dims: int, in_chans: int, out_chans: int, norm: str, dropout: float = 0.0, kernel_size: KernelSizeType = 3,
                 act_layer: type = nn.GELU) -> None:
        if n_dims not in {2, 3}:
            raise ValueError(f"Invalid n_dims, must be 2 or 3, got {n_dims}.")
        if not isinstance(kernel_size, int) and len(kernel_size) != n_dims:
            raise ValueError(f"Invalid kernel_size {kernel_size}, must be an integer or a tuple of {n_dims} integers.")
        super().__init__()
        if act_layer is not nn.GELU:
            raise NotImplementedError("cinema_amd ConvResBlock: GELU only (what every CineMA model uses).")
        conv_cls = Conv2d if n_dims == 2 else Conv3d
        self.norm1 = get_conv_norm(n_dims=n_dims, in_chans=in_chans, norm=norm)
        self.norm2 = get_conv_norm(n_dims=n_dims, in_chans=out_chans, norm=norm)
        self.conv1 = conv_cls(in_chans, out_chans, kernel_size=kernel_size, padding="same")
        self.conv2 = conv_cls(out_chans, out_chans, kernel_size=kernel_size, padding="same")
        self.dropout = nn.Dropout(dropout)
        self.act = act_layer()
        self.shortcut = conv_cls(in_chans, out_chans, kernel_size=1) if in_chans != out_chans else nn.Identity()

    def set_grad_ckpt(self, enable: bool = True) -> None:
        self.grad_ckpt = enable
        self.conv1.set_grad_ckpt(enable)
        self.conv2.set_grad_ckpt(enable)
        if hasattr(self.shortcut, "set_grad_ckpt"):
            self.shortcut.set_grad_ckpt(enable)

    def tape_forward(self, tp: T.Tape, x: Volume) -> Volume:
        """x: fp32 channels-last rows; returns fp32 rows with ``out_chans`` channels."""
        h = T.op_layernorm(tp, x.var, self.norm1.weight, self.norm1.bias, self.norm1.eps, act=1)
        h = T.op_conv_same(tp, h, x.batch, x.spatial, self.conv1.weight, self.conv1.bias, out_f32=True)
        h = T.op_layernorm(tp, h, self.norm2.weight, self.norm2.bias, self.norm2.eps, act=1)
        if self.training and self.dropout.p > 0:  # nn.Dropout between the activation and conv2 (conv.py:343)
            h = T.op_dropout(tp, h, float(self.dropout.p))
        if isinstance(self.shortcut, nn.Identity):
            res = x.var
        else:
            res = T.op_linear(tp, T.op_cast_bf16(tp, x.var), self.shortcut.weight, self.shortcut.bias, out_f32=True)
        y = T.op_conv_same(tp, h, x.batch, x.spatial, self.conv2.weight, self.conv2.bias, residual=res)
        return Volume(y, x.batch, x.spatial, self.conv2.out_channels)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Channels-first (b, C, *S) in/out like the reference."""
        b, c, *sp = x.shape
        rows = x.movedim(1, -1).contiguous().float().reshape(-1, c)

        def run(tp: T.Tape, xv: T.Var):  # noqa: ANN202
            return [self.tape_forward(tp, Volume(xv, b, tuple(sp), c)).var], []

        (y,) = T.taped_call(run, [rows], list(self.parameters()))
        return y.reshape(b, *sp, -1).movedim(-1, 1).contiguous()
