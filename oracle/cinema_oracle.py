"""CPU oracle for the CineMA MAE hot path -- TEST INFRASTRUCTURE ONLY.

A functional, fp32, CPU restatement of the algorithm of ``mathpluscode/CineMA``'s MAE
pre-training forward (``cinema/mae/mae.py:504-612``) written against a flat ``state_dict``
(the reference's own key names) instead of ``nn.Module`` objects.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file;
the product package ``cinema_amd`` never does (it fails loudly without its HIP library).

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function below against
golden vectors captured from the imported upstream reference (``oracle/make_golden.py``,
fixtures under ``tests/golden/``), including loss, predictions, metrics, gradients and a 3-step
AdamW trajectory.

The arithmetic itself lives in PyTorch ATen (``torch==2.11.0`` pinned upstream, 2.10.0 here) and
timm 1.0.15's ``Mlp`` (fc1 -> GELU -> fc2); both are restated with plain ``torch`` ops.

Every function cites the reference lines it follows.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F  # noqa: N812

Tensor = torch.Tensor
Params = dict  # flat {reference state_dict key: tensor}


# ----------------------------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------------------------
@dataclass
class MAEConfig:
    """Constructor arguments of the reference ``CineMA`` (``cinema/mae/mae.py:288-313``)."""

    image_size_dict: dict
    in_chans_dict: dict
    enc_patch_size_dict: dict
    enc_scale_factor_dict: dict
    enc_conv_chans: list
    enc_conv_n_blocks: int
    enc_embed_dim: int
    enc_depth: int
    enc_n_heads: int
    dec_embed_dim: int
    dec_depth: int
    dec_n_heads: int
    mlp_ratio: int = 4
    norm_target: bool = False
    cross_attn: bool = True
    norm_eps: float = 1e-5
    views: list = field(default_factory=list)

    def __post_init__(self) -> None:
        self.views = list(self.image_size_dict.keys())
        for d in (self.image_size_dict, self.enc_patch_size_dict, self.enc_scale_factor_dict):
            for k in d:
                d[k] = tuple(int(v) for v in d[k])

    # derived geometry ------------------------------------------------------------------------
    def patch_sizes(self, view: str) -> list:
        """``DownsampleEncoder.patch_sizes`` (``cinema/convvit.py:85``)."""
        return [self.enc_patch_size_dict[view]] + [self.enc_scale_factor_dict[view]] * len(self.enc_conv_chans)

    def grid_size(self, view: str) -> tuple:
        size = self.image_size_dict[view]
        for p in self.patch_sizes(view):
            size = tuple(s // q for s, q in zip(size, p))
        return size

    def dec_patch_size(self, view: str) -> tuple:
        """``get_decoder_patch_size`` (``cinema/mae/mae.py:207-228``)."""
        out = (1,) * len(self.image_size_dict[view])
        for p in self.patch_sizes(view):
            out = tuple(a * b for a, b in zip(out, p))
        return out


VIT_SIZES = {  # ``get_vit_config`` (``cinema/vit.py:784-831``)
    "tiny": dict(enc_embed_dim=16, enc_depth=1, enc_n_heads=2, dec_embed_dim=16, dec_depth=1, dec_n_heads=2),
    "base": dict(enc_embed_dim=768, enc_depth=12, enc_n_heads=12, dec_embed_dim=512, dec_depth=8, dec_n_heads=16),
    "large": dict(enc_embed_dim=1024, enc_depth=24, enc_n_heads=16, dec_embed_dim=512, dec_depth=8, dec_n_heads=16),
    "huge": dict(enc_embed_dim=1280, enc_depth=32, enc_n_heads=16, dec_embed_dim=512, dec_depth=8, dec_n_heads=16),
}


def mae_config(size: str, sax_size=None, lax_size=None, views=("sax", "lax_2c", "lax_3c", "lax_4c"),  # noqa: ANN001
               patch_size=(4, 4, 1), scale_factor=(2, 2, 1), conv_chans=(64, 128), conv_n_blocks=2, in_chans=1) -> MAEConfig:
    """Same mapping as the reference ``get_model`` (``cinema/mae/mae.py:231-282``)."""
    img, pch, scl, chn = {}, {}, {}, {}
    for v in views:
        nd = 3 if v == "sax" else 2
        img[v] = tuple(sax_size) if v == "sax" else tuple(lax_size)
        pch[v], scl[v], chn[v] = tuple(patch_size[:nd]), tuple(scale_factor[:nd]), in_chans
    return MAEConfig(image_size_dict=img, in_chans_dict=chn, enc_patch_size_dict=pch, enc_scale_factor_dict=scl,
                     enc_conv_chans=list(conv_chans), enc_conv_n_blocks=conv_n_blocks, **VIT_SIZES[size])


# ----------------------------------------------------------------------------------------------
# patches and positional tables
# ----------------------------------------------------------------------------------------------
def patchify(image: Tensor, patch_size: tuple) -> Tensor:
    """(b, C, *S) -> (b, prod(S/p), prod(p)*C), token raster order, feature order (p..., C).

    ``patchify_2d/3d/4d`` (``cinema/vit.py:67-161``); raises ``ValueError`` like the reference.
    """
    n = len(patch_size)
    if n not in (2, 3, 4):
        raise ValueError(f"Patchify only supports 2D, 3D, and 4D images, got {n}D.")
    b, c, *size = image.shape
    for s, p in zip(size, patch_size):
        if s % p != 0:
            raise ValueError(f"Input size ({s}) cannot be divided by patch size ({p}).")
    grid = [s // p for s, p in zip(size, patch_size)]
    x = image.reshape(b, c, *[v for gp in zip(grid, patch_size) for v in gp])
    grid_axes = [2 + 2 * i for i in range(n)]
    patch_axes = [3 + 2 * i for i in range(n)]
    x = x.permute(0, *grid_axes, *patch_axes, 1).contiguous()
    return x.reshape(b, math.prod(grid), math.prod(patch_size) * c)


def unpatchify(x: Tensor, patch_size: tuple, grid_size: tuple) -> Tensor:
    """Inverse of :func:`patchify` (``cinema/vit.py:164-256``)."""
    b, n_patches, chans = x.shape
    if n_patches != math.prod(grid_size):
        raise ValueError(f"Number of patches {n_patches} != product of grid size {grid_size}.")
    if chans % math.prod(patch_size) != 0:
        raise ValueError(f"Number of channels {chans} is not divisible by product of patch size {patch_size}.")
    if len(patch_size) != len(grid_size):
        raise ValueError(f"Patch size {patch_size} and grid size {grid_size} do not match.")
    n = len(patch_size)
    if n not in (2, 3, 4):
        raise ValueError(f"Unpatchify only supports 2D, 3D, and 4D images, got {n}D.")
    x = x.reshape(b, *grid_size, *patch_size, -1)
    order = [0, 1 + 2 * n]
    for i in range(n):
        order += [1 + i, 1 + n + i]
    x = x.permute(*order).contiguous()
    return x.reshape(b, -1, *[g * p for g, p in zip(grid_size, patch_size)])


def sincos_1d(dim: int, pos: np.ndarray, max_period: int = 10000) -> np.ndarray:
    """``get_1d_sincos_pos_embed_from_grid`` (``cinema/vit.py:347-383``): [sin | cos] halves."""
    if dim % 2 != 0:
        raise ValueError(f"Embedding dimension must be divisible by 2, got {dim}.")
    half = dim // 2
    omega = np.exp(-np.log(max_period) * np.arange(half, dtype=np.float32) / half)
    ang = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)


def sincos_pos_embed(dim: int, grid_size: tuple) -> Tensor:
    """(1, prod(grid), dim) table, ``get_pos_embed`` (``cinema/vit.py:386-443``).

    Faithful quirks: ``np.meshgrid`` default ``indexing='xy'`` (``vit.py:421``), per-axis width
    ``dim // n`` rounded down to even with zero padding (``vit.py:398-405``).
    """
    grid = np.stack(np.meshgrid(*[np.arange(s, dtype=np.float32) for s in grid_size]), axis=0)
    n = grid.shape[0]
    d = dim // n
    d -= d % 2
    emb = np.concatenate([sincos_1d(d, grid[i]) for i in range(n)], axis=1)
    if dim - d * n > 0:
        emb = np.concatenate([emb, np.zeros((emb.shape[0], dim - d * n))], axis=1)
    return torch.from_numpy(emb).float().unsqueeze(0)


def upsample_mask(mask: Tensor, scale_factor: tuple) -> Tensor:
    """Nearest-neighbour upsampling of a (b, *grid) bool mask (``cinema/convvit.py:24-51``)."""
    if mask.ndim != len(scale_factor) + 1:
        raise ValueError("mask must have the same number of dimensions as scale_factor except batch")
    for axis, f in enumerate(scale_factor):
        mask = mask.repeat_interleave(int(f), dim=axis + 1)
    return mask


def random_patch_mask(batch: int, n_patches: int, ratio: float, generator: torch.Generator | None = None) -> Tensor:
    """``get_batch_random_patch_mask`` (``cinema/mae/mae.py:30-65``): True = removed."""
    if ratio < 0:
        raise ValueError(f"mask_ratio must be positive, got {ratio}.")
    if ratio == 0:
        return torch.zeros(batch, n_patches, dtype=torch.bool)
    noise = torch.rand(batch, n_patches, generator=generator)
    rank = torch.argsort(torch.argsort(noise, dim=1), dim=1)
    return rank >= int(n_patches * (1 - ratio))


# ----------------------------------------------------------------------------------------------
# layers (functional, on a flat parameter dict)
# ----------------------------------------------------------------------------------------------
def _ln(x: Tensor, p: Params, key: str, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), p[f"{key}.weight"], p[f"{key}.bias"], eps)


def _lin(x: Tensor, p: Params, key: str) -> Tensor:
    return F.linear(x, p[f"{key}.weight"], p.get(f"{key}.bias"))


def _conv(x: Tensor, p: Params, key: str, stride=1, padding=0, groups: int = 1) -> Tensor:  # noqa: ANN001
    fn = F.conv3d if x.ndim == 5 else F.conv2d
    return fn(x, p[f"{key}.weight"], p.get(f"{key}.bias"), stride=stride, padding=padding, groups=groups)


def _conv_ln(x: Tensor, p: Params, key: str, eps: float = 1e-6) -> Tensor:
    """``ConvLayerNorm`` (``cinema/conv.py:169-187``): LN over the channel axis of (b, C, *S); eps 1e-6 (``conv.py:190``)."""
    return _ln(x.movedim(1, -1), p, key, eps).movedim(-1, 1).contiguous()


def attention(q_in: Tensor, k_in: Tensor, p: Params, key: str, n_heads: int) -> Tensor:
    """``Attention.forward`` (``cinema/vit.py:482-522``), explicit softmax branch (``:513-517``).

    ``kv`` out-features are laid out (2, heads, head_dim) (``vit.py:499``).
    """
    b, tq, c = q_in.shape
    tk = k_in.shape[1]
    hd = c // n_heads
    q = _lin(q_in, p, f"{key}.q").reshape(b, tq, n_heads, hd).permute(0, 2, 1, 3)
    kv = _lin(k_in, p, f"{key}.kv").reshape(b, tk, 2, n_heads, hd).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    w = torch.softmax((q * hd**-0.5) @ k.transpose(-2, -1), dim=-1)
    o = (w @ v).transpose(1, 2).reshape(b, tq, c)
    return _lin(o, p, f"{key}.proj")


def block(q: Tensor, k: Tensor | None, p: Params, key: str, n_heads: int, eps: float) -> Tensor:
    """Pre-LN ``Block`` (``cinema/vit.py:587-609``); cross-attention keys are *not* normed (``:589``)."""
    qn = _ln(q, p, f"{key}.norm1", eps)
    q = q + attention(qn, qn if k is None else k, p, f"{key}.attn", n_heads)
    h = F.gelu(_lin(_ln(q, p, f"{key}.norm2", eps), p, f"{key}.mlp.fc1"))
    return q + _lin(h, p, f"{key}.mlp.fc2")


def masked_conv_block(x: Tensor, vis: Tensor | None, p: Params, key: str) -> Tensor:
    """``MaskedConvBlock.forward`` (``cinema/conv.py:400-415``); ``vis`` is (b, *S), 1 = visible."""
    h = _conv(_conv_ln(x, p, f"{key}.norm1"), p, f"{key}.conv1")
    if vis is not None:
        h = vis.unsqueeze(1).to(h.dtype) * h
    h = _conv(h, p, f"{key}.dw_conv", padding=2, groups=h.shape[1])  # kernel 5 in every axis (conv.py:385)
    x = x + _conv(h, p, f"{key}.conv2")
    h = F.gelu(_conv(_conv_ln(x, p, f"{key}.norm2"), p, f"{key}.mlp.fc1"))
    return x + _conv(h, p, f"{key}.mlp.fc2")


def downsample_encoder(image: Tensor, mask: Tensor | None, p: Params, key: str, cfg: MAEConfig, view: str):  # noqa: ANN201
    """``DownsampleEncoder.forward`` (``cinema/convvit.py:165-207``) -> (skips, tokens (b, n, E))."""
    sizes = cfg.patch_sizes(view)
    b = image.shape[0]
    eff = [math.prod(s[i] for s in sizes) for i in range(len(sizes[0]))]
    grid = tuple(s // e for s, e in zip(image.shape[2:], eff))
    vis_masks: list = [None] * len(cfg.enc_conv_chans)
    if mask is not None:
        m = mask.reshape(b, *grid)
        for lvl in range(len(sizes) - 1, 0, -1):  # coarse -> fine (convvit.py:186-192)
            m = upsample_mask(m, sizes[lvl])
            vis_masks[lvl - 1] = ~m
    skips, x = [], image
    for lvl in range(len(cfg.enc_conv_chans)):
        kb = f"{key}.conv_blocks.{lvl}"
        x = F.gelu(_conv_ln(_conv(x, p, f"{kb}.patch_embed.conv", stride=sizes[lvl]), p, f"{kb}.patch_embed.norm"))
        for j in range(cfg.enc_conv_n_blocks):
            x = masked_conv_block(x, vis_masks[lvl], p, f"{kb}.conv.{j}")
        skips.append(x)
    tok = _lin(_lin(patchify(x, sizes[-1]), p, f"{key}.patch_embed.proj"), p, f"{key}.linear")
    pe = p[f"{key}.pos_embed"]
    if grid != cfg.grid_size(view):  # ``interpolate_pos_encoding`` (convvit.py:140-163)
        e = pe.shape[-1]
        mode = {2: "bicubic", 3: "trilinear"}[len(grid)]
        pe = F.interpolate(pe.reshape(1, *cfg.grid_size(view), e).movedim(-1, 1), size=grid, mode=mode)
        pe = pe.movedim(1, -1).reshape(1, -1, e)
    return skips, tok + pe


def multi_scale_fusion(skips: list, x: Tensor, mask: Tensor | None, p: Params, key: str, eps: float) -> Tensor:
    """``MultiScaleFusion.forward`` (``cinema/convvit.py:265-291``): LN(x + sum_i down_i(skip_i)[kept])."""
    for i, skip in enumerate(skips):
        w = p[f"{key}.down_convs.{i}.weight"]
        down = _conv(skip, p, f"{key}.down_convs.{i}", stride=tuple(w.shape[2:])).flatten(2).transpose(1, 2)
        if mask is not None:
            down = down[~mask].reshape(x.shape[0], -1, x.shape[-1])
        x = x + down
    return _ln(x, p, f"{key}.norm", eps)


def mse_loss(target: Tensor, pred: Tensor, mask: Tensor, norm_target: bool, epsilon: float = 1e-6):  # noqa: ANN201
    """``mse_loss`` (``cinema/mae/mae.py:107-152``): masked-patch MSE + metrics (unbiased variance)."""
    mean = target.mean(dim=-1, keepdim=True)
    std = target.var(dim=-1, keepdim=True) ** 0.5
    metrics = {"target_mean": mean.mean(), "target_std": std.mean()}
    if norm_target:
        target = (target - mean) / (std + epsilon)
    target = target[mask].reshape(pred.shape)
    loss = ((pred - target.detach()) ** 2).mean()
    metrics["mse_loss"] = loss
    if norm_target and target.shape[1] > 0:
        metrics["normed_target_max"] = target.max()
        metrics["pred_max"] = pred.max()
    return loss, metrics


# ----------------------------------------------------------------------------------------------
# the model
# ----------------------------------------------------------------------------------------------
def encode(p: Params, cfg: MAEConfig, image_dict: dict, mask_dict: dict | None):  # noqa: ANN201
    """Stem + token selection + ViT encoder + fusion (``cinema/mae/mae.py:541-563`` / ``:478-499``)."""
    views = list(image_dict)
    if any(v not in cfg.views for v in views):
        raise ValueError(f"views {views} must be in self.input_keys {cfg.views}.")
    b = image_dict[views[0]].shape[0]
    toks, skips_all, n_keep = [], [], []
    for v in views:
        m = None if mask_dict is None else mask_dict[v]
        skips, x = downsample_encoder(image_dict[v], m, p, f"enc_down_dict.{v}", cfg, v)
        if m is not None:
            x = x[~m].reshape(b, -1, x.shape[-1])  # raster order of kept tokens (mae.py:550)
        toks.append(x)
        skips_all.append(skips)
        n_keep.append(x.shape[1])
    x = torch.cat([p["encoder.cls_token"].expand(b, -1, -1), *toks], dim=1)  # vit.py:672-674
    for i in range(cfg.enc_depth):
        x = block(x, None, p, f"encoder.blocks.{i}", cfg.enc_n_heads, cfg.norm_eps)
    x = _ln(x, p, "encoder.norm", cfg.norm_eps)
    parts = list(torch.split(x, [1, *n_keep], dim=1))
    for i, v in enumerate(views):
        m = None if mask_dict is None else mask_dict[v]
        parts[i + 1] = multi_scale_fusion(skips_all[i], parts[i + 1], m, p, f"enc_fusion_dict.{v}", cfg.norm_eps)
    return parts, n_keep


def feature_forward(p: Params, cfg: MAEConfig, image_dict: dict) -> dict:
    """``CineMA.feature_forward`` (``cinema/mae/mae.py:457-502``)."""
    parts, _ = encode(p, cfg, image_dict, None)
    return dict(zip(["cls", *image_dict], parts))


def convvit_features(p: Params, cfg: MAEConfig, image_dict: dict, mask_dict: dict | None) -> dict:
    """``ConvViT.feature_forward`` (``cinema/convvit.py:459-503``): every token is embedded, ``mask_dict`` only masks the conv stem,
    the fusion runs with ``mask=None``."""
    views = list(image_dict)
    if any(v not in cfg.views for v in views):
        raise ValueError(f"views {views} must be in self.input_keys {cfg.views}.")
    b = image_dict[views[0]].shape[0]
    toks, skips_all, n_tok = [], [], []
    for v in views:
        m = None if mask_dict is None else mask_dict[v]
        skips, x = downsample_encoder(image_dict[v], m, p, f"enc_down_dict.{v}", cfg, v)
        toks.append(x)
        skips_all.append(skips)
        n_tok.append(x.shape[1])
    x = torch.cat([p["encoder.cls_token"].expand(b, -1, -1), *toks], dim=1)
    for i in range(cfg.enc_depth):
        x = block(x, None, p, f"encoder.blocks.{i}", cfg.enc_n_heads, cfg.norm_eps)
    x = _ln(x, p, "encoder.norm", cfg.norm_eps)
    parts = list(torch.split(x, [1, *n_tok], dim=1))
    out = {"cls": parts[0]}
    for i, v in enumerate(views):
        out[v] = multi_scale_fusion(skips_all[i], parts[i + 1], None, p, f"enc_fusion_dict.{v}", cfg.norm_eps)
    return out


def convvit_forward(p: Params, cfg: MAEConfig, image_dict: dict, mask_dict: dict | None = None, reduce: str = "all") -> Tensor:
    """``ConvViT.forward`` (``cinema/convvit.py:505-558``): per-view head on the token mean (+ the cls head), averaged."""
    x = convvit_features(p, cfg, image_dict, mask_dict)
    if reduce == "cls":
        return _lin(x["cls"], p, "pred_head_dict.cls")[:, 0]
    if reduce not in {"patch", "all"}:
        raise NotImplementedError(f"Unsupported reduce method {reduce}.")
    outs = [_lin(x[v].mean(dim=1, keepdim=True), p, f"pred_head_dict.{v}") for v in cfg.views]
    if reduce == "all":
        outs.append(_lin(x["cls"], p, "pred_head_dict.cls"))
    return torch.cat(outs, dim=1).mean(dim=1)


def conv_res_block(x: Tensor, p: Params, key: str) -> Tensor:
    """``ConvResBlock.forward`` (``cinema/conv.py:328-348``), dropout = 0: norm-GELU-conv-norm-GELU-conv + (1x1 conv | identity) shortcut."""
    pad = tuple(k // 2 for k in p[f"{key}.conv1.weight"].shape[2:])
    h = _conv(F.gelu(_conv_ln(x, p, f"{key}.norm1")), p, f"{key}.conv1", padding=pad)
    h = _conv(F.gelu(_conv_ln(h, p, f"{key}.norm2")), p, f"{key}.conv2", padding=pad)
    return h + (_conv(x, p, f"{key}.shortcut") if f"{key}.shortcut.weight" in p else x)


def _conv_transpose(x: Tensor, p: Params, key: str) -> Tensor:
    w = p[f"{key}.weight"]
    fn = F.conv_transpose3d if x.ndim == 5 else F.conv_transpose2d
    return fn(x, w, p.get(f"{key}.bias"), stride=tuple(w.shape[2:]))


def upsample_decoder(embeddings: list, p: Params, key: str, n_levels: int, n_blocks: int = 2) -> Tensor:
    """``UpsampleDecoder.forward`` (``cinema/segmentation/convunetr.py:89-106``)."""
    embeddings = list(embeddings)
    x = embeddings.pop()
    for i in range(n_levels):
        x = _conv_transpose(x, p, f"{key}.blocks.{i}.up")
        skip = embeddings.pop()
        if skip is not None:
            x = x + skip
        for j in range(n_blocks):
            x = conv_res_block(x, p, f"{key}.blocks.{i}.conv.{j}")
    return x


def convunetr_forward(p: Params, cfg: MAEConfig, dec_chans: tuple, n_layers_wo_skip: int, n_downsample_layers: int, image_dict: dict) -> dict:
    """``ConvUNetR.forward`` (``cinema/segmentation/convunetr.py:422-485``): logits (b, out_chans, *image_size) per view."""
    views = list(image_dict)
    if any(v not in cfg.views for v in views):
        raise ValueError(f"views {views} must be in self.input_keys {cfg.views}.")
    b = image_dict[views[0]].shape[0]
    toks, skips_all, n_tok = [], [], []
    for v in views:
        skips, x = downsample_encoder(image_dict[v], None, p, f"enc_down_dict.{v}", cfg, v)
        toks.append(x)
        skips_all.append(skips)
        n_tok.append(x.shape[1])
    x = torch.cat([p["encoder.cls_token"].expand(b, -1, -1), *toks], dim=1)
    for i in range(cfg.enc_depth):
        x = block(x, None, p, f"encoder.blocks.{i}", cfg.enc_n_heads, cfg.norm_eps)
    x = _ln(x, p, "encoder.norm", cfg.norm_eps)
    parts = torch.split(x, [1, *n_tok], dim=1)[1:]
    preds = {}
    for i, v in enumerate(views):
        grid = tuple(s // e for s, e in zip(image_dict[v].shape[2:], [math.prod(q[d] for q in cfg.patch_sizes(v)) for d in range(len(cfg.patch_sizes(v)[0]))]))
        xv = parts[i].permute(0, 2, 1).reshape(b, -1, *grid)
        skips_view = list(skips_all[i]) + [xv]
        for j in range(n_downsample_layers):
            w = p[f"dec_down_blocks_dict.{v}.{j}.weight"]
            xv = _conv(xv, p, f"dec_down_blocks_dict.{v}.{j}", stride=tuple(w.shape[2:]))
            skips_view.append(xv)
        emb = [conv_res_block(image_dict[v], p, f"dec_image_conv_block_dict.{v}")] + [None] * n_layers_wo_skip
        for j in range(len(skips_view)):
            emb.append(conv_res_block(skips_view[j], p, f"dec_conv_blocks_dict.{v}.{j}"))
        y = upsample_decoder(emb, p, f"decoder_dict.{v}", len(dec_chans))
        preds[v] = _conv(y, p, f"pred_head_dict.{v}")
    return preds


def mae_forward(p: Params, cfg: MAEConfig, image_dict: dict, mask_dict: dict):  # noqa: ANN201
    """``CineMA.forward`` (``cinema/mae/mae.py:504-612``) with the random masks injected.

    Returns (loss, pred_dict, metrics).
    """
    views = list(image_dict)
    parts, n_keep = encode(p, cfg, image_dict, mask_dict)
    b = parts[0].shape[0]
    x = _lin(torch.cat(parts, dim=1), p, "dec_linear")  # mae.py:567
    parts = torch.split(x, [1, *n_keep], dim=1)
    vis, msk, n_masked = [], [], []
    for i, v in enumerate(views):  # ``DecoderEmbedding`` (mae.py:92-104,179-204)
        pe = p[f"dec_embed_dict.{v}.pos_embed"].expand(b, -1, -1)
        m = mask_dict[v]
        e = pe.shape[-1]
        vis.append(parts[i + 1] + pe[~m].reshape(b, -1, e))
        msk.append(p[f"dec_embed_dict.{v}.mask_token"] + pe[m].reshape(b, -1, e))
        n_masked.append(msk[-1].shape[1])
    if cfg.cross_attn:  # mae.py:579-582
        x_q, x_k = torch.cat([parts[0], *msk], dim=1), torch.cat(vis, dim=1)
    else:  # mae.py:584-585
        x_q, x_k = torch.cat([parts[0], *vis, *msk], dim=1), None
    for i in range(cfg.dec_depth):
        x_q = block(x_q, x_k, p, f"decoder.blocks.{i}", cfg.dec_n_heads, cfg.norm_eps)
    x = _ln(x_q[:, x_q.shape[1] - sum(n_masked):], p, "decoder.norm", 1e-5)  # vit.py:738 default eps
    outs = torch.split(x, n_masked, dim=1)
    preds, losses, metrics = {}, [], {}
    for i, v in enumerate(views):
        preds[v] = _lin(outs[i], p, f"pred_head_dict.{v}")
        lv, mv = mse_loss(patchify(image_dict[v], cfg.dec_patch_size(v)), preds[v], mask_dict[v], cfg.norm_target)
        metrics.update({f"{v}_{k}": val for k, val in mv.items()})
        if torch.isfinite(lv):
            losses.append(lv)
    loss = sum(losses) / len(losses) if losses else torch.tensor(float("nan"))
    metrics["loss"] = loss
    return loss, preds, metrics


# ----------------------------------------------------------------------------------------------
# parameters and the optimisation step (harness semantics of cinema/mae/pretrain.py + cinema/optim.py)
# ----------------------------------------------------------------------------------------------
def param_shapes(cfg: MAEConfig) -> dict:
    """{key: (shape, kind)} for every ``state_dict`` entry of the reference ``CineMA``."""
    out: dict = {}

    def lin(key: str, i: int, o: int) -> None:
        out[f"{key}.weight"], out[f"{key}.bias"] = ((o, i), "xavier"), ((o,), "zero")

    def ln(key: str, c: int) -> None:
        out[f"{key}.weight"], out[f"{key}.bias"] = ((c,), "one"), ((c,), "zero")

    def conv(key: str, i: int, o: int, k: tuple, groups: int = 1) -> None:
        out[f"{key}.weight"], out[f"{key}.bias"] = ((o, i // groups, *k), "conv"), ((o,), "conv_bias")

    def vit_block(key: str, c: int) -> None:
        ln(f"{key}.norm1", c)
        lin(f"{key}.attn.q", c, c)
        lin(f"{key}.attn.kv", c, 2 * c)
        lin(f"{key}.attn.proj", c, c)
        ln(f"{key}.norm2", c)
        lin(f"{key}.mlp.fc1", c, c * cfg.mlp_ratio)
        lin(f"{key}.mlp.fc2", c * cfg.mlp_ratio, c)

    e, d = cfg.enc_embed_dim, cfg.dec_embed_dim
    for v in cfg.views:
        nd = len(cfg.image_size_dict[v])
        sizes, grid = cfg.patch_sizes(v), cfg.grid_size(v)
        kd = f"enc_down_dict.{v}"
        out[f"{kd}.pos_embed"] = ((1, math.prod(grid), e), "sincos")
        cin = cfg.in_chans_dict[v]
        for lvl, ch in enumerate(cfg.enc_conv_chans):
            kb = f"{kd}.conv_blocks.{lvl}"
            conv(f"{kb}.patch_embed.conv", cin, ch, sizes[lvl])
            ln(f"{kb}.patch_embed.norm", ch)
            for j in range(cfg.enc_conv_n_blocks):
                kc = f"{kb}.conv.{j}"
                ln(f"{kc}.norm1", ch)
                ln(f"{kc}.norm2", ch)
                conv(f"{kc}.conv1", ch, ch, (1,) * nd)
                conv(f"{kc}.conv2", ch, ch, (1,) * nd)
                conv(f"{kc}.dw_conv", ch, ch, (5,) * nd, groups=ch)
                conv(f"{kc}.mlp.fc1", ch, 4 * ch, (1,) * nd)
                conv(f"{kc}.mlp.fc2", 4 * ch, ch, (1,) * nd)
            cin = ch
        lin(f"{kd}.patch_embed.proj", cin * math.prod(sizes[-1]), e)
        lin(f"{kd}.linear", e, e)
    for v in cfg.views:
        sizes, grid = cfg.patch_sizes(v), cfg.grid_size(v)
        size = cfg.image_size_dict[v]
        for lvl, ch in enumerate(cfg.enc_conv_chans):
            size = tuple(s // q for s, q in zip(size, sizes[lvl]))
            conv(f"enc_fusion_dict.{v}.down_convs.{lvl}", ch, e, tuple(s // g for s, g in zip(size, grid)))
        ln(f"enc_fusion_dict.{v}.norm", e)
    out["encoder.cls_token"] = ((1, 1, e), "token")
    for i in range(cfg.enc_depth):
        vit_block(f"encoder.blocks.{i}", e)
    ln("encoder.norm", e)
    lin("dec_linear", e, d)
    for v in cfg.views:
        out[f"dec_embed_dict.{v}.pos_embed"] = ((1, math.prod(cfg.grid_size(v)), d), "sincos")
        out[f"dec_embed_dict.{v}.mask_token"] = ((1, 1, d), "token")
    for i in range(cfg.dec_depth):
        vit_block(f"decoder.blocks.{i}", d)
    ln("decoder.norm", d)
    for v in cfg.views:
        lin(f"pred_head_dict.{v}", d, math.prod(cfg.dec_patch_size(v)) * cfg.in_chans_dict[v])
    return out


def init_params(cfg: MAEConfig, seed: int = 0) -> Params:
    """Random parameters with the reference's *distributions* (``cinema/vit.py:32-64``): xavier-uniform
    Linear, zero bias, N(0, .02) tokens, torch-default conv init, frozen sin-cos tables.  (The draw
    order differs from the reference's module construction order; tests that need bit-identical
    weights load a ``state_dict``.)"""
    g = torch.Generator().manual_seed(seed)
    p: Params = {}
    shapes = param_shapes(cfg)
    for key, (shape, kind) in shapes.items():
        if kind == "xavier":
            bound = math.sqrt(6.0 / (shape[0] + shape[1]))
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif kind == "zero":
            t = torch.zeros(shape)
        elif kind == "one":
            t = torch.ones(shape)
        elif kind == "token":
            t = torch.randn(shape, generator=g) * 0.02
        elif kind in ("conv", "conv_bias"):
            wshape = shape if kind == "conv" else shapes[key.replace(".bias", ".weight")][0]
            bound = 1.0 / math.sqrt(math.prod(wshape[1:]))
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif kind == "sincos":
            view = key.split(".")[1]
            t = sincos_pos_embed(shape[-1], cfg.grid_size(view))
        else:
            raise AssertionError(kind)
        p[key] = t
    return p


def trainable_keys(p: Params) -> list:
    """Everything except the frozen sin-cos tables (``requires_grad=False``, ``cinema/vit.py:441``)."""
    return [k for k in p if not k.endswith("pos_embed")]


def weight_decay_groups(p: Params, weight_decay: float) -> list:
    """timm ``param_groups_weight_decay`` as called at ``cinema/mae/pretrain.py:365``:
    ``ndim <= 1`` or ``*.bias`` -> no decay; tokens (ndim 3) ARE decayed."""
    no_decay = [k for k in trainable_keys(p) if p[k].ndim <= 1 or k.endswith(".bias")]
    decay = [k for k in trainable_keys(p) if k not in set(no_decay)]
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


def lr_at(step: float, warmup_steps: float, max_n_steps: float, lr: float, min_lr: float) -> float:
    """``adjust_learning_rate`` (``cinema/optim.py:21-52``): linear warm-up then half cosine."""
    if step < warmup_steps:
        return lr * step / warmup_steps
    return min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * (step - warmup_steps) / (max_n_steps - warmup_steps)))


class Trainer:
    """fp32 CPU restatement of one optimisation step of ``pretrain_one_epoch``
    (``cinema/mae/pretrain.py:242-269``) + ``GradScaler.__call__`` (``cinema/optim.py:204-215``):
    forward, backward, global-L2 clip at ``clip_grad`` (pre-clip norm returned), AdamW."""

    def __init__(self, params: Params, cfg: MAEConfig, lr: float = 1e-3, betas=(0.9, 0.95), weight_decay: float = 0.05,  # noqa: ANN001
                 clip_grad: float | None = 5.0) -> None:
        self.cfg = cfg
        self.p = {k: v.detach().clone().float() for k, v in params.items()}
        for k in trainable_keys(self.p):
            self.p[k].requires_grad_(True)
        groups = [{"params": [self.p[k] for k in g["params"]], "weight_decay": g["weight_decay"]}
                  for g in weight_decay_groups(self.p, weight_decay)]
        self.opt = torch.optim.AdamW(groups, lr=lr, betas=tuple(betas))
        self.clip_grad = clip_grad

    def set_lr(self, lr: float) -> None:
        for g in self.opt.param_groups:
            g["lr"] = lr * g.get("lr_scale", 1.0)

    def step(self, image_dict: dict, mask_dict: dict):  # noqa: ANN201
        loss, preds, metrics = mae_forward(self.p, self.cfg, image_dict, mask_dict)
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        tensors = [self.p[k] for k in trainable_keys(self.p)]
        if self.clip_grad is not None:
            norm = torch.nn.utils.clip_grad_norm_(tensors, self.clip_grad)
        else:
            norm = torch.linalg.vector_norm(torch.stack([t.grad.norm() for t in tensors if t.grad is not None]))
        self.opt.step()
        return loss.detach(), norm.detach(), preds, metrics


def segmentation_loss_one_view(logits: Tensor, labels: Tensor):  # noqa: ANN201
    """``_segmentation_loss`` (``cinema/segmentation/train.py:77-103``) with monai's ``one_hot`` / ``DiceLoss(include_background=False,
    softmax=True)`` (monai 1.5.2, absent here) restated: smooth_nr = smooth_dr = 1e-5, sums over the spatial axes per (sample, class),
    mean over samples x foreground classes.  Returns (loss, metrics).

    The Dice term: monai==1.5.2 (reference pyproject.toml:20) is not installed here and the reference holds no test or golden value for this
    function, so it cannot be pinned against the library itself.  It is pinned instead (round 6) against an INDEPENDENT second statement of monai's
    published ``DiceLoss.forward`` (softmax, drop channel 0, per-item spatial sums, ``1 - (2 I + smooth_nr) / (G + P + smooth_dr)``, mean) written in
    another style - float64 loops over samples / classes / voxels, ``oracle/second_opinion.py`` - on the committed vectors
    ``tests/golden/second_opinion.safetensors`` (five random cases, an absent class, ignored voxels, an all-background volume, 2-D), plus the
    hand-computed known answer of the tests."""
    labels = labels.long()
    c = logits.shape[1]
    target = F.one_hot(labels.clamp(min=0).squeeze(1), c).movedim(-1, 1).to(logits.dtype)
    ce = F.cross_entropy(logits, labels.squeeze(1), ignore_index=-1)
    prob = torch.softmax(logits, dim=1)[:, 1:]
    tgt = target[:, 1:]
    axes = tuple(range(2, logits.ndim))
    inter = (prob * tgt).sum(axes)
    den = tgt.sum(axes) + prob.sum(axes)
    dice = (1.0 - (2.0 * inter + 1e-5) / (den + 1e-5)).mean()
    loss = dice + ce
    return loss, {"cross_entropy": ce, "mean_dice_loss": dice, "loss": loss}


# ----------------------------------------------------------------------------------------------
# evaluation path of the segmentation task (cinema/segmentation/train.py:148-286, cinema/transform.py:13-124, cinema/metric.py:21-96)
# ----------------------------------------------------------------------------------------------
def patch_grid(image_size: tuple, patch_size: tuple, patch_overlap: tuple) -> np.ndarray:
    """``get_patch_grid`` (``cinema/transform.py:13-50``): per axis arange(0, size - patch + 1, patch - overlap) plus a last window flush with
    the border; the grid is the 'ij' product.  PINNED by the reference's own known answers (``cinema/transform_test.py:13-115``) and by
    ``tests/golden/seg_eval.safetensors``."""
    axes = []
    for size, patch, overlap in zip(image_size, patch_size, patch_overlap):
        if patch > size:
            raise ValueError(f"Patch size {patch} should be <= image size {size}.")
        starts = list(range(0, size - patch + 1, patch - overlap))
        if starts[-1] != size - patch:
            starts.append(size - patch)
        axes.append(starts)
    grid = [()]
    for starts in axes:
        grid = [g + (s,) for g in grid for s in starts]
    return np.array(grid, dtype=np.int64).reshape(-1, len(image_size))


def sliding_window_logits(forward, image_dict: dict, patch_size_dict: dict) -> dict:  # noqa: ANN001
    """``segmentation_forward`` (``cinema/segmentation/train.py:148-221``) around any ``forward(image_dict) -> logits_dict``: one window at a
    time over the single view that needs patching (half-patch overlap), per-window softmax, overlap mean (``aggregate_patches``,
    ``cinema/transform.py:86-124``), log; the other views: log of the mean window probability.  PINNED by ``tests/golden/seg_eval.safetensors``."""
    views = list(image_dict)
    for v, image in image_dict.items():
        if any(s < p for s, p in zip(image.shape[2:], patch_size_dict[v])):
            raise ValueError(f"For view {v}, image size {image.shape[2:]} is smaller than patch size {patch_size_dict[v]}.")
    need = {v: tuple(image_dict[v].shape[2:]) != tuple(patch_size_dict[v]) for v in views}
    if not any(need.values()):
        return forward(image_dict)
    if sum(need.values()) > 1:
        raise ValueError(f"Only support patching on one view for now, but got {need}.")
    if image_dict[views[0]].shape[0] != 1:
        raise ValueError(f"Expected batch size 1 for patching, but got {image_dict[views[0]].shape[0]}.")
    vp = next(v for v in views if need[v])
    image = image_dict[vp][0]
    size, patch = tuple(image.shape[1:]), tuple(patch_size_dict[vp])
    starts = patch_grid(size, patch, tuple(s // 2 for s in patch))
    outs = {v: [] for v in views}
    for st in starts:
        sl = (slice(None),) + tuple(slice(int(a), int(a) + p) for a, p in zip(st, patch))
        res = forward({v: image[sl][None] if v == vp else image_dict[v] for v in views})
        for v in views:
            outs[v].append(res[v])
    result = {}
    for v in views:
        prob = torch.softmax(torch.cat(outs[v], dim=0), dim=1)
        if v == vp:
            acc = torch.zeros((prob.shape[1], *size), dtype=prob.dtype)
            cnt = torch.zeros(size, dtype=torch.float32)
            for i, st in enumerate(starts):
                sl = tuple(slice(int(a), int(a) + p) for a, p in zip(st, patch))
                acc[(slice(None), *sl)] += prob[i]
                cnt[sl] += 1
            result[v] = torch.log(acc / cnt[None])[None]
        else:
            result[v] = torch.log(prob.mean(dim=0))[None]
    return result


def _iou_ignore_empty(y_pred: Tensor, y: Tensor) -> Tensor:
    """monai 1.5.2 ``compute_iou(y_pred, y)`` with its defaults (include_background=True, ignore_empty=True): per (batch, channel)
    |y & y_pred| / (|y| + |y_pred| - |y & y_pred|), NaN where y is empty.  PINNED through ``stability_score`` by the reference's known answers
    (``cinema/metric_test.py:60-102``)."""
    axes = tuple(range(2, y.ndim))
    inter = (y * y_pred).sum(axes).float()
    y_o, p_o = y.sum(axes).float(), y_pred.sum(axes).float()
    return torch.where(y_o > 0, inter / (y_o + p_o - inter), torch.full_like(inter, float("nan")))


def stability_score(logits: Tensor, threshold: float = 0.0, threshold_offset: float = 1.0) -> Tensor:
    """``cinema/metric.py:21-45``: IoU between the masks (logits - class mean) >= threshold + offset and >= threshold - offset."""
    norm = logits - logits.mean(dim=1, keepdim=True)
    return _iou_ignore_empty((norm >= threshold + threshold_offset).long(), (norm >= threshold - threshold_offset).long())


def hausdorff_distance_95(pred_label: Tensor, true_label: Tensor, n_classes: int, spacing: tuple, percentile: float = 95.0) -> Tensor:
    """monai 1.5.2 ``compute_hausdorff_distance(y_pred, y, include_background=False, percentile=95, spacing=spacing)`` as the reference calls it
    (``cinema/segmentation/train.py:262-267``), restated from the published algorithm with the scipy functions monai itself uses on CPU (PARITY-UNPINNED:
    monai is absent and the reference holds no value): per (sample, foreground class) the surfaces are ``mask ^ binary_erosion(mask)``
    (``get_mask_edges``; scipy's default cross structure, border value 0), the directed distances ``distance_transform_edt(~other_surface, sampling=
    spacing)[this_surface]`` (``get_surface_distance``), the result the maximum over the two directions of the ``percentile`` quantile (``torch.quantile``,
    linear interpolation).  An empty surface on exactly one side fills the distance map with inf -> inf; both empty -> NaN.
    label maps (batch, *spatial) -> (batch, n_classes)."""
    from scipy import ndimage

    nd = pred_label.dim() - 1
    sp = [float(v) for v in spacing][:nd]
    out = torch.full((pred_label.shape[0], n_classes), float("nan"))
    for b in range(pred_label.shape[0]):
        for k in range(1, n_classes + 1):
            p, t = (pred_label[b] == k).numpy(), (true_label[b] == k).numpy()
            ep, et = ndimage.binary_erosion(p) ^ p, ndimage.binary_erosion(t) ^ t
            if not ep.any() and not et.any():
                continue
            if not ep.any() or not et.any():
                out[b, k - 1] = float("inf")
                continue
            d_pt = torch.from_numpy(ndimage.distance_transform_edt(~et, sampling=sp)[ep]).float()
            d_tp = torch.from_numpy(ndimage.distance_transform_edt(~ep, sampling=sp)[et]).float()
            out[b, k - 1] = torch.maximum(torch.quantile(d_pt, percentile / 100.0), torch.quantile(d_tp, percentile / 100.0))
    return out


def segmentation_metrics(logits: Tensor, labels: Tensor, spacing: tuple) -> dict:
    """``segmentation_metrics`` (``cinema/segmentation/train.py:224-286``): argmax one-hot prediction, monai
    ``compute_dice`` (2 |A & B| / (|A| + |B|), NaN where the ground truth is empty: ``ignore_empty=True``) and ``compute_iou``, stability score,
    volumes in ml (``cinema/metric.py:84-96``).  ``compute_dice`` is PARITY-UNPINNED (monai absent, no reference value); it is cross-checked in
    the tests against an independent float64 count-based derivation."""
    n_classes = logits.shape[1] - 1
    lab = labels.squeeze(1).long()
    pred = F.one_hot(torch.argmax(logits, dim=1), n_classes + 1).movedim(-1, 1)
    true = F.one_hot(lab, n_classes + 1).movedim(-1, 1)
    axes = tuple(range(2, pred.ndim))
    inter = (pred * true).sum(axes).float()
    t_o, p_o = true.sum(axes).float(), pred.sum(axes).float()
    dice = torch.where(t_o > 0, 2.0 * inter / (t_o + p_o), torch.full_like(inter, float("nan")))
    iou = _iou_ignore_empty(pred, true)
    stab = stability_score(logits)
    vox = float(np.prod(spacing)) / 1000.0
    out = {}
    for i in range(n_classes):
        k = i + 1
        out[f"class_{k}_dice_score"], out[f"class_{k}_iou_score"], out[f"class_{k}_stability_score"] = dice[:, k], iou[:, k], stab[:, k]
        out[f"class_{k}_true_volume"], out[f"class_{k}_pred_volume"] = t_o[:, k] * vox, p_o[:, k] * vox
    out["mean_dice_score"], out["mean_iou_score"] = dice[:, 1:].mean(-1), iou[:, 1:].mean(-1)
    out["mean_stability_score"] = stab[:, 1:].mean(-1)
    hd = hausdorff_distance_95(torch.argmax(logits, dim=1), lab, n_classes, spacing)
    for i in range(n_classes):
        out[f"class_{i + 1}_hausdorff_distance_95"] = hd[:, i]
    out["mean_hausdorff_distance_95"] = hd.mean(-1)
    return out


# ----------------------------------------------------------------------------------------------
# input transforms of the pre-training loader (cinema/mae/pretrain.py:157-200) -- monai 1.5.2 restated (monai absent, no reference value test: pinned
# since round 6 against the independent loop-style statement oracle/second_opinion.py on tests/golden/second_opinion.safetensors): Zoom(keep_size=True, padding_mode="constant") = interpolate(scale_factor, recompute_scale_factor=True, align_corners
# False) + centred pad / crop; ScaleIntensity(minv=0, maxv=1) = (x - min) / (max - min), zeros for a constant image; SpatialPad(method="end").
# ----------------------------------------------------------------------------------------------
def input_transform(x: Tensor, zoom: float, padded_size: tuple, cubic: bool) -> Tensor:
    """x (*size) fp32 -> (*padded_size)."""
    size = tuple(x.shape)
    if zoom != 1.0:
        mode = "bicubic" if cubic else ("trilinear" if x.dim() == 3 else "bilinear")
        y = F.interpolate(x[None, None], scale_factor=[float(zoom)] * x.dim(), mode=mode, align_corners=False, recompute_scale_factor=True)[0, 0]
        out = torch.zeros(size, dtype=x.dtype)
        src, dst = [], []
        for s, o in zip(size, y.shape):
            if o < s:
                b = (s - o) // 2
                src.append(slice(0, o)); dst.append(slice(b, b + o))
            else:
                b = (o - s) // 2
                src.append(slice(b, b + s)); dst.append(slice(0, s))
        out[tuple(dst)] = y[tuple(src)]
        x = out
    mn, mx = x.min(), x.max()
    x = (x - mn) / (mx - mn) if float(mx) > float(mn) else torch.zeros_like(x)
    pad = []
    for s, p in zip(reversed(size), reversed(tuple(padded_size))):
        pad += [0, p - s]
    return F.pad(x, pad)


# ---------------------------------------------------------------------------------------------------------------------
# ConvViT fine-tuning heads: losses and patch-averaged evaluation forward
# ---------------------------------------------------------------------------------------------------------------------
def classification_loss_value(logits: Tensor, labels: Tensor, label_smoothing: float = 0.1) -> Tensor:
    """``F.cross_entropy(logits, label, label_smoothing=eps)`` as ``classification_loss`` calls it (``cinema/classification/train.py:104-108``), written out:
    mean_i [(1 - eps) * (-log p_i[y_i]) + eps / c * sum_j (-log p_i[j])]."""
    logp = logits.double() - torch.logsumexp(logits.double(), dim=1, keepdim=True)
    nll = -logp.gather(1, labels.long().reshape(-1, 1))[:, 0]
    smooth = -logp.mean(dim=1)
    return ((1.0 - label_smoothing) * nll + label_smoothing * smooth).mean().float()


def regression_loss_values(preds: Tensor, label: Tensor) -> dict:
    """The values of ``regression_loss`` (``cinema/regression/train.py:40-55``): ``F.mse_loss`` (the loss), ``F.l1_loss`` and the label / prediction ranges."""
    d = preds.double() - label.double()
    return {"mse_loss": float((d * d).mean()), "loss": float((d * d).mean()), "mae_loss": float(d.abs().mean()), "max_label": float(label.max()),
            "min_label": float(label.min()), "max_pred": float(preds.max()), "min_pred": float(preds.min())}


def patch_average_forward(forward, image_dict: dict, patch_size_dict: dict, task: str) -> Tensor:  # noqa: ANN001
    """``classification_forward`` / ``regression_forward`` (``cinema/classification/train.py:113-178``, ``cinema/regression/train.py:59-123``): ``forward`` maps an
    image dict to (batch, n).  One over-sized view (batch 1) is cut into half-overlapping patches (``patch_grid`` above, i.e. ``get_patch_grid``); classification
    averages the per-patch softmax and returns its log, regression averages the predictions."""
    views = list(image_dict)
    need = {v: tuple(image_dict[v].shape[2:]) != tuple(patch_size_dict[v]) for v in views}
    if not any(need.values()):
        return forward(image_dict)
    if sum(need.values()) > 1 or image_dict[views[0]].shape[0] != 1:
        raise ValueError("one over-sized view and batch size 1")
    v0 = next(v for v in views if need[v])
    img, ps = image_dict[v0][0], tuple(patch_size_dict[v0])
    starts = patch_grid(tuple(img.shape[1:]), ps, tuple(s // 2 for s in ps))
    outs = []
    for st in starts:
        sl = (slice(None),) + tuple(slice(int(a), int(a) + b) for a, b in zip(st, ps))
        outs.append(forward({v: img[sl][None] if v == v0 else image_dict[v] for v in views}))
    out = torch.cat(outs, dim=0)
    if task == "classification":
        return torch.log(torch.softmax(out, dim=1).mean(dim=0, keepdim=True))
    return out.mean(dim=0, keepdim=True)
