"""ViT layers on the HIP tape (interface of the reference ``cinema/vit.py``).

Same class / attribute / parameter names as the reference so that ``state_dict`` keys and seeded construction match.
Compute per block (``Block.tape_forward``): LayerNorm kernel -> fused q|k|v MFMA GEMM -> flash attention ->
proj GEMM with fused bias + fp32 residual -> LayerNorm -> fc1 GEMM with fused bias + exact GELU -> fc2 GEMM with fused
bias + residual.  The residual stream stays fp32; everything that feeds an MFMA is bf16 (the reference runs the same
split under ``torch.autocast(bf16)``, ``cinema/mae/pretrain.py:251``).
"""

from __future__ import annotations

import math

import numpy as np
import torch
from torch import nn

from cinema_amd import tape as T
from cinema_amd.conv import Linear, _CkptFlag
from cinema_amd.rotary import RotaryEmbedding


def init_weights(m: nn.Module) -> None:
    """Xavier-uniform Linear weights, zero biases, unit LayerNorm (reference ``cinema/vit.py:32-48``)."""
    if isinstance(m, nn.Linear):
        torch.nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
        if m.weight is not None:
            nn.init.constant_(m.weight, 1.0)


def get_tokens(embed_dim: int, n_tokens: int) -> nn.Parameter:
    """Learnable (1, n_tokens, embed_dim) token, N(0, 0.02) (reference ``cinema/vit.py:51-64``)."""
    token = nn.Parameter(torch.zeros(1, n_tokens, embed_dim))
    nn.init.normal_(token, std=0.02)
    return token


# ---------------------------------------------------------------------------------------------------------------
# patchify / unpatchify: pure index permutations kept as tensor views (used by callers to rebuild images,
# e.g. examples/inference/mae.py); the training path gathers patches inside the HIP kernels instead.
# ---------------------------------------------------------------------------------------------------------------
def patchify(image: torch.Tensor, patch_size: tuple) -> torch.Tensor:
    """(batch, C, *S) -> (batch, n_patches, prod(patch)*C); token raster order, features ordered (patch..., C).

    Same contract and errors as the reference dispatcher (``cinema/vit.py:67-161``).
    """
    n = len(patch_size)
    if n not in (2, 3, 4):
        raise ValueError(f"Patchify only supports 2D, 3D, and 4D images, got {n}D.")
    batch, chans, *size = image.shape
    names = ("height", "width", "depth", "time")
    for i, (s, p) in enumerate(zip(size, patch_size)):
        if s % p != 0:
            raise ValueError(f"Input {names[i]} ({s}) cannot be divided by patch size ({p}).")
    grid = [s // p for s, p in zip(size, patch_size)]
    x = image.reshape(batch, chans, *[v for pair in zip(grid, patch_size) for v in pair])
    x = x.permute(0, *[2 + 2 * i for i in range(n)], *[3 + 2 * i for i in range(n)], 1).contiguous()
    return x.reshape(batch, math.prod(grid), math.prod(patch_size) * chans)


def unpatchify(x: torch.Tensor, patch_size: tuple, grid_size: tuple) -> torch.Tensor:
    """Inverse of :func:`patchify` (reference ``cinema/vit.py:227-256``)."""
    batch, n_patches, chans = x.shape
    if n_patches != math.prod(grid_size):
        raise ValueError(f"Number of patches {n_patches} != product of grid size {math.prod(grid_size)} for {grid_size}.")
    if chans % math.prod(patch_size) != 0:
        raise ValueError(f"Number of channels {chans} is not divisible by product of patch size {math.prod(patch_size)} for {patch_size}.")
    if len(patch_size) != len(grid_size):
        raise ValueError(f"Patch size {patch_size} and grid size {grid_size} do not match.")
    n = len(patch_size)
    if n not in (2, 3, 4):
        raise ValueError(f"Unpatchify only supports 2D, 3D, and 4D images, got {n}D.")
    x = x.reshape(batch, *grid_size, *patch_size, -1)
    order = [0, 1 + 2 * n]
    for i in range(n):
        order += [1 + i, 1 + n + i]
    return x.permute(*order).contiguous().reshape(batch, -1, *[g * p for g, p in zip(grid_size, patch_size)])


def _patchify_nd(n: int):  # noqa: ANN202
    def fn(image: torch.Tensor, patch_size: tuple) -> torch.Tensor:
        if len(patch_size) != n or image.dim() != n + 2:
            raise ValueError(f"patchify_{n}d expects a (batch, C, *{n} spatial axes) image and a {n}-tuple patch size, got {tuple(image.shape)} and {patch_size}.")
        return patchify(image, patch_size)

    fn.__name__ = fn.__qualname__ = f"patchify_{n}d"
    fn.__doc__ = f"{n}-D case of :func:`patchify` (reference ``cinema/vit.py:67-142``): same token / feature order and the same errors."
    return fn


def _unpatchify_nd(n: int):  # noqa: ANN202
    def fn(x: torch.Tensor, patch_size: tuple, grid_size: tuple) -> torch.Tensor:
        if len(patch_size) != n or len(grid_size) != n:
            raise ValueError(f"unpatchify_{n}d expects {n}-tuples for patch_size and grid_size, got {patch_size} and {grid_size}.")
        return unpatchify(x, patch_size, grid_size)

    fn.__name__ = fn.__qualname__ = f"unpatchify_{n}d"
    fn.__doc__ = f"{n}-D case of :func:`unpatchify` (reference ``cinema/vit.py:164-225``)."
    return fn


patchify_2d, patchify_3d, patchify_4d = _patchify_nd(2), _patchify_nd(3), _patchify_nd(4)
unpatchify_2d, unpatchify_3d, unpatchify_4d = _unpatchify_nd(2), _unpatchify_nd(3), _unpatchify_nd(4)


# ---------------------------------------------------------------------------------------------------------------
# frozen sin-cos positional tables (reference cinema/vit.py:347-443), built once on the host
# ---------------------------------------------------------------------------------------------------------------
def get_1d_sincos_pos_embed_from_grid(embed_dim: int, grid: np.ndarray, max_period: int = 10000, dtype: type = np.float32) -> np.ndarray:
    if embed_dim % 2 != 0:
        raise ValueError(f"Embedding dimension must be divisible by 2, got {embed_dim}.")
    half = embed_dim // 2
    omega = np.exp(-np.log(max_period) * np.arange(half, dtype=dtype) / half)
    angles = np.einsum("m,d->md", grid.reshape(-1), omega)
    return np.concatenate([np.sin(angles), np.cos(angles)], axis=1)


def get_nd_sincos_pos_embed_from_grid(embed_dim: int, grid: np.ndarray) -> np.ndarray:
    """(M, embed_dim) table of a position grid of shape (n, ...): each of the n axes gets an even width ``embed_dim // n`` (rounded down to even), the rest is
    zero padding (reference ``cinema/vit.py:386-405``)."""
    n = grid.shape[0]
    width = embed_dim // n
    width -= width % 2
    emb = np.concatenate([get_1d_sincos_pos_embed_from_grid(width, grid[i]) for i in range(n)], axis=1)
    pad = embed_dim - width * n
    if pad > 0:
        emb = np.concatenate([emb, np.zeros((emb.shape[0], pad))], axis=1)
    return emb


def get_nd_sincos_pos_embed(embed_dim: int, grid_size: tuple) -> np.ndarray:
    """(prod(grid), embed_dim).  Keeps the reference's ``np.meshgrid`` 'xy' axis order (``vit.py:421``) and its even-width /
    zero-padding rule for dimensions that do not divide evenly (``vit.py:398-405``)."""
    grid = np.stack(np.meshgrid(*[np.arange(s, dtype=np.float32) for s in grid_size]), axis=0)
    return get_nd_sincos_pos_embed_from_grid(embed_dim, grid)


def get_pos_embed(embed_dim: int, grid_size: tuple) -> nn.Parameter:
    """Frozen (1, prod(grid), embed_dim) parameter -- it is part of the ``state_dict`` like in the reference (``vit.py:426-443``)."""
    table = torch.from_numpy(get_nd_sincos_pos_embed(embed_dim, grid_size)).float().unsqueeze(0)
    return nn.Parameter(table, requires_grad=False)


class PatchEmbed(nn.Module, _CkptFlag):
    """patchify + Linear (reference ``cinema/vit.py:259-344``).  Inside the models the gather and the GEMM are HIP kernels
    restricted to the kept tokens; this class owns ``proj`` and the geometry attributes the callers read."""

    def __init__(self, image_size: tuple, patch_size: tuple, in_chans: int, embed_dim: int, norm_layer: type | None = None, bias: bool = True,
                 strict_image_size: bool = False, dynamic_img_pad: bool = False) -> None:
        super().__init__()
        self.n_dims = len(image_size)
        self.patch_size = tuple(patch_size)
        self.image_size = tuple(image_size)
        self.grid_size = tuple(s // p for s, p in zip(self.image_size, self.patch_size))
        self.n_patches = math.prod(self.grid_size)
        self.strict_image_size = strict_image_size
        self.dynamic_img_pad = dynamic_img_pad
        self.proj = Linear(in_features=in_chans * math.prod(patch_size), out_features=embed_dim, bias=bias)
        torch.nn.init.xavier_uniform_(self.proj.weight.data.view([self.proj.weight.shape[0], -1]))
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()
        if norm_layer:
            raise NotImplementedError("PatchEmbed(norm_layer=...) is unused by the CineMA models.")

    def set_grad_ckpt(self, enable: bool = True) -> None:
        self.grad_ckpt = enable
        self.proj.set_grad_ckpt(enable)

    def check_size(self, image_size: tuple) -> None:
        """Input-size validation of the reference forward (``vit.py:318-331``)."""
        if self.strict_image_size:
            for i in range(self.n_dims):
                if self.image_size[i] != image_size[i]:
                    raise ValueError(f"Input size ({image_size}) doesn't match config (batch, channel) + {self.image_size}.")
        elif not self.dynamic_img_pad:
            for i in range(self.n_dims):
                if image_size[i] % self.patch_size[i] != 0:
                    raise ValueError(f"Input size ({image_size}) should be divisible by patch size ({self.patch_size}).")

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        size = tuple(image.shape[2:])
        self.check_size(size)
        if self.dynamic_img_pad:
            # ``vit.py:332-337``: the (0, missing) pairs are listed in AXIS order and handed to ``F.pad``, which applies its first pair to the LAST axis - kept as
            # it is (the amounts land on the right axes only when they agree, e.g. an isotropic patch on a cubic image)
            pad = sum([(0, (p - size[i] % p) % p) for i, p in enumerate(self.patch_size)], ())
            image = torch.nn.functional.pad(image, pad)
        return self.proj(patchify(image, self.patch_size))


class Attention(nn.Module):
    """Multi-head attention with separate ``q`` and fused ``kv`` projections (reference ``cinema/vit.py:446-522``)."""

    def __init__(self, dim: int, n_heads: int = 8, qkv_bias: bool = False, qk_norm: bool = False, attn_drop: float = 0.0, proj_drop: float = 0.0,
                 norm_layer: type = nn.LayerNorm, norm_eps: float = 1e-5, rotary: bool = False) -> None:
        super().__init__()
        if dim % n_heads != 0:
            raise ValueError(f"dim {dim} should be divisible by n_heads {n_heads}")
        if attn_drop > 0.0:
            raise NotImplementedError("attention-probability dropout (attn_drop) is not used by the CineMA models and has no HIP path.")
        if qk_norm and rotary:
            raise NotImplementedError("qk_norm together with rotary embedding has no HIP path (no CineMA model uses either with the other).")
        self.n_heads = n_heads
        self.head_dim = dim // n_heads
        self.scale = self.head_dim**-0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        # qk_norm (``vit.py:476-477``): LayerNorm over the head dimension of q and k; runs unfused (tape.op_attention on given projections)
        self.q_norm = norm_layer(self.head_dim, eps=norm_eps) if qk_norm else nn.Identity()
        self.k_norm = norm_layer(self.head_dim, eps=norm_eps) if qk_norm else nn.Identity()
        self.qk_norm = bool(qk_norm)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        # The reference hands RotaryEmbedding q, k as (batch, heads, tokens, head_dim) (vit.py:496-499), so its table is indexed by the HEAD
        # and q, k of one head get the same rotation, which cancels in q.k^T up to rounding (golden "rotary/out" ==
        # "rotary/out_plain" to 6e-8).  The HIP path applies that same head-indexed rotation (cinema_rope_heads) to the fused q|k rows.
        self.rotary = RotaryEmbedding(self.head_dim) if rotary else None

    def tape_forward(self, tp: T.Tape, xq: T.Var, xk: T.Var | None, batch: int, shared_kv: tuple | None = None) -> T.Var:
        """xq: bf16 [b*tq, c] (normed queries); xk: bf16 [b*tk, c] or None for self-attention.  Returns bf16 [b*tq, c].
        ``shared_kv`` = (tape.SharedKV, index): k|v of this block were projected together with the other decoder blocks'."""
        if xk is not None and self.rotary:
            raise ValueError("Rotary positional embedding is not supported with different query and key.")
        if self.qk_norm:
            if shared_kv is not None:
                raise NotImplementedError("qk_norm with the shared decoder k|v projection")
            c, hd = self.n_heads * self.head_dim, self.head_dim
            q = T.op_linear(tp, xq, self.q.weight, self.q.bias)
            k, v = T.op_split_cols(tp, T.op_linear(tp, xq if xk is None else xk, self.kv.weight, self.kv.bias), [c, c])
            qn = T.op_layernorm(tp, T.op_view(tp, q, (q.data.shape[0] * self.n_heads, hd)), self.q_norm.weight, self.q_norm.bias, self.q_norm.eps)
            kn = T.op_layernorm(tp, T.op_view(tp, k, (k.data.shape[0] * self.n_heads, hd)), self.k_norm.weight, self.k_norm.bias, self.k_norm.eps)
            return T.op_attention(tp, T.op_view(tp, qn, tuple(q.data.shape)), T.op_view(tp, kn, tuple(k.data.shape)), v, batch, self.n_heads)
        if xk is None:
            rope = None
            if self.rotary is not None:
                dev = xq.data.device
                rope = T.const(("rope_tables", self.n_heads, self.head_dim, self.rotary.base, self.rotary.scaling_factor, str(dev)),
                               lambda: self.rotary.head_tables(self.n_heads, dev))
            return T.op_self_attention(tp, xq, batch, self.n_heads, self.q.weight, self.q.bias, self.kv.weight, self.kv.bias, rope=rope, fp8=T.FP8_FORWARD)
        return T.op_cross_attention(tp, xq, xk, batch, self.n_heads, self.q.weight, self.q.bias, self.kv.weight, self.kv.bias, fp8=T.FP8_FORWARD,
                                    shared=shared_kv)

    def forward(self, q: torch.Tensor, k: torch.Tensor | None = None) -> torch.Tensor:
        b, tq, c = q.shape

        def run(tp: T.Tape, qv: T.Var, kv: T.Var | None = None):  # noqa: ANN202
            q16 = T.op_cast_bf16(tp, qv)
            k16 = None if kv is None else T.op_cast_bf16(tp, kv)
            o = self.tape_forward(tp, q16, k16, b)
            p = self.proj_drop.p if self.training else 0.0
            if p > 0.0:
                T.begin_stochastic(self, q.device)
                return [T.op_cast_f32(tp, T.op_dropout(tp, T.op_linear(tp, o, self.proj.weight, self.proj.bias), p))], []
            return [T.op_linear(tp, o, self.proj.weight, self.proj.bias, out_f32=True)], []

        inputs = [q.float().reshape(-1, c).contiguous()] + ([] if k is None else [k.float().reshape(-1, c).contiguous()])
        (y,) = T.taped_call(run, inputs, list(self.parameters()))
        return y.reshape(b, tq, c)


class Mlp(nn.Module):
    """fc1 -> GELU -> fc2 parameter container with timm's attribute names (timm 1.0.15 ``Mlp``, used at ``cinema/vit.py:570-575``)."""

    def __init__(self, in_features: int, hidden_features: int | None = None, out_features: int | None = None, act_layer: type = nn.GELU,
                 norm_layer: type | None = None, bias: bool = True, drop: float = 0.0, use_conv: bool = False) -> None:
        super().__init__()
        if act_layer is not nn.GELU or norm_layer is not None or use_conv:
            raise NotImplementedError("cinema_amd Mlp: GELU, no inner norm, Linear layers.")
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.norm = nn.Identity()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
        self.drop2 = nn.Dropout(drop)


class DropPath(nn.Module):
    """Stochastic depth per sample (timm 1.0.15 ``DropPath``, ``scale_by_keep=True``; used at ``cinema/vit.py:561,577``).  Inside ``Block`` the
    draw and the scaled residual add are HIP launches on the tape; this module holds the rate and serves direct calls on fp32 GPU tensors."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True) -> None:
        super().__init__()
        if not scale_by_keep:
            raise NotImplementedError("DropPath(scale_by_keep=False) is not used by the CineMA models.")
        self.drop_prob = float(drop_prob)
        self.scale_by_keep = scale_by_keep

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.drop_prob == 0.0 or not self.training:
            return x
        from cinema_amd import hip as K

        b = x.shape[0]
        rows = x.float().reshape(b, -1).contiguous()
        K.rng_advance(x.device)
        scale = K.droppath_scale(b, self.drop_prob, 0x5EED, x.device)
        pad = (-rows.shape[1]) % 4
        if pad:
            raise NotImplementedError("DropPath direct call: the per-sample element count must be a multiple of 4")
        return K.scale_rows_add(rows, scale, 1).reshape(x.shape).to(x.dtype)

    def extra_repr(self) -> str:
        return f"drop_prob={round(self.drop_prob, 3):0.3f}"


class LayerScale(nn.Module):
    """timm 1.0.15 ``LayerScale`` (``timm/models/vision_transformer.py``; used at ``cinema/vit.py:561,576``): x * gamma with gamma = init_values * ones(dim)."""

    def __init__(self, dim: int, init_values: float = 1e-5, inplace: bool = False) -> None:
        super().__init__()
        self.inplace = inplace
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        c = x.shape[-1]
        (y,) = T.taped_call(lambda tp, xv: ([T.op_layerscale(tp, xv, self.gamma)], []), [x.float().reshape(-1, c).contiguous()], [self.gamma])
        return y.reshape(x.shape).to(x.dtype)


class Block(nn.Module, _CkptFlag):
    """Pre-LN transformer block with optional cross-attention keys (reference ``cinema/vit.py:525-609``)."""

    def __init__(self, dim: int, n_heads: int, mlp_ratio: int, norm_layer: type, norm_eps: float, drop_path: float, qkv_bias: bool, rotary: bool,
                 act_layer: type, mlp_layer: type, qk_norm: bool = False, proj_drop: float = 0.0, attn_drop: float = 0.0,
                 init_values: float | None = None) -> None:
        super().__init__()
        # stochastic depth (the fine-tuning configs use 0.1, cinema/segmentation/acdc/config.yaml:65): identity in eval mode like timm's DropPath;
        # in training mode the residual adds go through tape.op_droppath_add (per-sample Philox draws, cinema_droppath_scale)
        self.drop_path_rate = float(drop_path)
        if mlp_layer is not Mlp and getattr(mlp_layer, "__name__", "") != "Mlp":
            raise NotImplementedError("only the GELU Mlp has a HIP path (SwiGLU is unused by the reference configs).")
        self.norm1 = norm_layer(dim, eps=norm_eps)
        self.attn = Attention(dim, n_heads=n_heads, qkv_bias=qkv_bias, qk_norm=qk_norm, attn_drop=attn_drop, proj_drop=proj_drop,
                              norm_layer=norm_layer, norm_eps=norm_eps, rotary=rotary)
        # LayerScale (init_values) and proj_drop (nn.Dropout behind the attention projection and inside the Mlp) take the block off the fused epilogues: the
        # residual adds, the scale and the dropout are launches of their own (tape/ops_options.py); no shipped configuration sets either
        self.ls1 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.proj_drop = float(proj_drop)
        self.drop_path1 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim, eps=norm_eps)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=proj_drop)
        self.ls2 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path2 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()

    def _param_list(self) -> list:
        pl = self.__dict__.get("_plist")
        if pl is None:
            pl = self.__dict__["_plist"] = [p for p in self.parameters() if p.requires_grad]
        return pl

    def tape_forward(self, tp: T.Tape, xq: T.Var, xk: T.Var | None, batch: int, shared_kv: tuple | None = None) -> T.Var:
        """xq: fp32 residual stream [b*tq, c]; xk: bf16 un-normed keys [b*tk, c] or None (``vit.py:589``)."""
        drop = self.drop_path_rate if self.training else 0.0
        # gradient all-reduce of this block may start once its backward ops are launched (k|v projected for all decoder blocks at once: those two parameters get
        # their gradients from op_shared_kv's backward, which marks them itself)
        pl = self._param_list()
        if shared_kv is not None:
            pl = [p for p in pl if p is not self.attn.kv.weight and p is not self.attn.kv.bias]
        T.mark_params(tp, pl)
        T.wgrad_group(tp)                      # ... which includes the grouped weight-gradient launch: its flush runs before that marker fires
        qn = T.op_layernorm(tp, xq, self.norm1.weight, self.norm1.bias, self.norm1.eps, fp8=True)
        att = self.attn.tape_forward(tp, qn, xk, batch, shared_kv=shared_kv)
        p_drop = self.proj_drop if self.training else 0.0
        if isinstance(self.ls1, LayerScale) or p_drop > 0.0:
            y = self._tape_forward_options(tp, xq, att, batch, drop, p_drop)
        elif drop > 0.0:  # q + drop_path1(path1(q)), q + drop_path2(path2(q)) (vit.py:606-609): the residual adds leave the GEMM epilogues
            h1 = T.op_linear(tp, att, self.attn.proj.weight, self.attn.proj.bias, out_f32=True, fp8=T.FP8_FORWARD)
            x1 = T.op_droppath_add(tp, h1, xq, batch, drop)
            xn2 = T.op_layernorm(tp, x1, self.norm2.weight, self.norm2.bias, self.norm2.eps, fp8=True)
            h2 = T.op_mlp(tp, xn2, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias, residual=None, fp8=T.FP8_FORWARD)
            y = T.op_droppath_add(tp, h2, x1, batch, drop)
        else:
            x1 = T.op_linear(tp, att, self.attn.proj.weight, self.attn.proj.bias, residual=xq, fp8=T.FP8_FORWARD)
            xn2 = T.op_layernorm(tp, x1, self.norm2.weight, self.norm2.bias, self.norm2.eps, fp8=True)
            y = T.op_mlp(tp, xn2, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias, residual=x1, fp8=T.FP8_FORWARD)
        T.wgrad_group_end(tp)
        return y

    def _tape_forward_options(self, tp: T.Tape, xq: T.Var, att: T.Var, batch: int, drop: float, p_drop: float) -> T.Var:
        """``q + drop_path1(ls1(proj_drop(proj(att))))`` then ``x1 + drop_path2(ls2(mlp(norm2(x1))))`` with timm's Mlp = fc1 -> GELU -> drop -> fc2 -> drop
        (``vit.py:593-609``), every piece a launch of its own."""
        ls = isinstance(self.ls1, LayerScale)

        def branch_out(h: T.Var, scale: nn.Module) -> T.Var:  # h: bf16 behind a dropout, fp32 otherwise -> fp32 branch output
            if ls:
                return T.op_layerscale(tp, h, scale.gamma)
            return h if h.data.dtype == torch.float32 else T.op_cast_f32(tp, h)

        def add(h: T.Var, res: T.Var) -> T.Var:
            return T.op_droppath_add(tp, h, res, batch, drop) if drop > 0.0 else T.op_add(tp, h, res, batch)

        if p_drop > 0.0:
            h1 = T.op_dropout(tp, T.op_linear(tp, att, self.attn.proj.weight, self.attn.proj.bias), p_drop)
        else:
            h1 = T.op_linear(tp, att, self.attn.proj.weight, self.attn.proj.bias, out_f32=True)
        x1 = add(branch_out(h1, self.ls1), xq)
        xn2 = T.op_layernorm(tp, x1, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        if p_drop > 0.0:
            a = T.op_dropout(tp, T.op_linear_gelu(tp, xn2, self.mlp.fc1.weight, self.mlp.fc1.bias), p_drop)
            h2 = T.op_dropout(tp, T.op_linear(tp, a, self.mlp.fc2.weight, self.mlp.fc2.bias), p_drop)
        else:
            h2 = T.op_mlp(tp, xn2, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias, residual=None)
        return add(branch_out(h2, self.ls2), x1)

    def forward(self, q: torch.Tensor, k: torch.Tensor | None = None) -> torch.Tensor:
        b, tq, c = q.shape

        def run(tp: T.Tape, qv: T.Var, kv: T.Var | None = None):  # noqa: ANN202
            if self.training and (self.proj_drop > 0.0 or self.drop_path_rate > 0.0):
                T.begin_stochastic(self, q.device)
            return [self.tape_forward(tp, qv, None if kv is None else T.op_cast_bf16(tp, kv), b)], []

        inputs = [q.float().reshape(-1, c).contiguous()] + ([] if k is None else [k.float().reshape(-1, c).contiguous()])
        (y,) = T.taped_call(run, inputs, list(self.parameters()))
        return y.reshape(b, tq, c)


def _make_blocks(embed_dim, depth, n_heads, mlp_ratio, qkv_bias, norm_layer, norm_eps, rotary, act_layer, mlp_layer, drop_path):  # noqa: ANN001, ANN202
    return nn.ModuleList([
        Block(dim=embed_dim, n_heads=n_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, norm_layer=norm_layer, norm_eps=norm_eps, rotary=rotary,
              act_layer=act_layer, mlp_layer=mlp_layer, drop_path=drop_path) for _ in range(depth)])


class ViTEncoder(nn.Module, _CkptFlag):
    """cls token + blocks + final norm (reference ``cinema/vit.py:612-698``)."""

    def __init__(self, embed_dim: int, depth: int, n_heads: int, mlp_ratio: int, qkv_bias: bool, norm_layer: type, norm_eps: float, rotary: bool,
                 act_layer: type, mlp_layer: type, drop_path: float) -> None:
        super().__init__()
        self.cls_token = get_tokens(embed_dim=embed_dim, n_tokens=1)
        self.blocks = _make_blocks(embed_dim, depth, n_heads, mlp_ratio, qkv_bias, norm_layer, norm_eps, rotary, act_layer, mlp_layer, drop_path)
        self.norm = norm_layer(embed_dim, eps=norm_eps)
        self.apply(init_weights)

    def set_grad_ckpt(self, enable: bool = True) -> None:
        self.grad_ckpt = enable
        for blk in self.blocks:
            blk.set_grad_ckpt(enable)

    def tape_forward(self, tp: T.Tape, x: T.Var, batch: int, collect: list | None = None) -> T.Var:
        """x: fp32 [b*(1+n), c] with the cls row already assembled in front.  Returns the normed fp32 sequence."""
        for i, blk in enumerate(self.blocks):
            x = blk.tape_forward(tp, x, None, batch)
            if collect is not None and i != len(self.blocks) - 1:
                collect.append(x)
        x = T.op_layernorm(tp, x, self.norm.weight, self.norm.bias, self.norm.eps, out_f32=True)
        if collect is not None:
            collect.append(x)
        return x

    def _assemble(self, tp: T.Tape, xv: T.Var, batch: int, n: int, c: int) -> T.Var:
        dev = xv.data.device
        ar = torch.arange(batch, dtype=torch.int32, device=dev)
        tok_dst = (ar[:, None] * (n + 1) + 1 + torch.arange(n, dtype=torch.int32, device=dev)[None]).reshape(-1)
        segs = [T.Segment(ar * (n + 1), src=self.cls_token), T.Segment(tok_dst, src=xv)]
        return T.op_assemble(tp, batch * (n + 1), c, segs, dev)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        b, n, c = x.shape

        def run(tp: T.Tape, xv: T.Var):  # noqa: ANN202
            return [self.tape_forward(tp, self._assemble(tp, xv, b, n, c), b)], []

        (y,) = T.taped_call(run, [x.float().reshape(-1, c).contiguous()], list(self.parameters()))
        return y.reshape(b, n + 1, c)

    def feature_forward(self, x: torch.Tensor) -> torch.Tensor:
        """(batch, 1+n, c, n_layers): every block output, the last one normed (reference ``vit.py:680-698``)."""
        b, n, c = x.shape

        def run(tp: T.Tape, xv: T.Var):  # noqa: ANN202
            feats: list = []
            self.tape_forward(tp, self._assemble(tp, xv, b, n, c), b, collect=feats)
            return feats, []

        outs = T.taped_call(run, [x.float().reshape(-1, c).contiguous()], list(self.parameters()))
        return torch.stack([o.reshape(b, n + 1, c) for o in outs], dim=-1)


class ViTDecoder(nn.Module, _CkptFlag):
    """Decoder blocks with optional cross-attention keys + final norm on the masked tokens (reference ``cinema/vit.py:701-781``)."""

    def __init__(self, embed_dim: int, depth: int, n_heads: int, mlp_ratio: int, qkv_bias: bool, norm_layer: type, norm_eps: float, rotary: bool,
                 act_layer: type, mlp_layer: type, drop_path: float) -> None:
        super().__init__()
        self.blocks = _make_blocks(embed_dim, depth, n_heads, mlp_ratio, qkv_bias, norm_layer, norm_eps, rotary, act_layer, mlp_layer, drop_path)
        self.norm = norm_layer(embed_dim)  # default eps, as the reference (vit.py:738)
        self.apply(init_weights)

    def set_grad_ckpt(self, enable: bool = True) -> None:
        self.grad_ckpt = enable
        for blk in self.blocks:
            blk.set_grad_ckpt(enable)

    def tape_forward(self, tp: T.Tape, xq: T.Var, xk: T.Var | None, batch: int) -> T.Var:
        """Runs the blocks and the final LayerNorm on every query row (the callers slice the masked rows; the norm is
        row-wise so normalising the extra cls/visible rows changes nothing).  Output bf16 [b*tq, c]."""
        attns = [blk.attn for blk in self.blocks]
        shared = T.op_shared_kv(tp, xk, attns) if (xk is not None and T.share_kv_ok(xk, attns)) else None
        for i, blk in enumerate(self.blocks):
            xq = blk.tape_forward(tp, xq, xk, batch, shared_kv=None if shared is None else (shared, i))
        return T.op_layernorm(tp, xq, self.norm.weight, self.norm.bias, self.norm.eps)

    def forward(self, x_q: torch.Tensor, x_k: torch.Tensor | None, n_enc_masked: int) -> torch.Tensor:
        b, tq, c = x_q.shape

        def run(tp: T.Tape, qv: T.Var, kv: T.Var | None = None):  # noqa: ANN202
            out = self.tape_forward(tp, qv, None if kv is None else T.op_cast_bf16(tp, kv), b)
            return [T.op_cast_f32(tp, out)], []

        inputs = [x_q.float().reshape(-1, c).contiguous()] + ([] if x_k is None else [x_k.float().reshape(-1, c).contiguous()])
        (y,) = T.taped_call(run, inputs, list(self.parameters()))
        return y.reshape(b, tq, c)[:, tq - n_enc_masked:, :]


def get_vit_config(size: str) -> dict:
    """tiny / base / large / huge (reference ``cinema/vit.py:784-831``)."""
    table = {
        "tiny": (16, 1, 2, 16, 1, 2),
        "base": (768, 12, 12, 512, 8, 16),
        "large": (1024, 24, 16, 512, 8, 16),
        "huge": (1280, 32, 16, 512, 8, 16),
    }
    if size not in table:
        raise ValueError(f"size must be in ['tiny', 'base', 'large', 'huge'], got {size}.")
    keys = ("enc_embed_dim", "enc_depth", "enc_n_heads", "dec_embed_dim", "dec_depth", "dec_n_heads")
    return dict(zip(keys, table[size]))
