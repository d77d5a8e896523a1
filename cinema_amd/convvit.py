"""Conv stem and multi-scale fusion on the HIP tape (interface of the reference ``cinema/convvit.py:24-291``).

In an MAE step (a mask with dropped tokens) the stem runs on the VISIBLE voxels only (``CINEMA_DENSE_STEM=1`` forces the
dense path): the reference evaluates every voxel and then reads the kept tokens (``cinema/mae/mae.py:548-550``); every stem op
except the depthwise conv is per-voxel and the depthwise conv input is zero at masked voxels (``cinema/conv.py:405-411``), so
the kept tokens, the loss and every gradient are the same numbers - 4x fewer rows for 75 % masking.

``DownsampleEncoder`` runs the stem on channels-last rows (fp32 residual stream, bf16 GEMM operands) and then embeds
ONLY the kept tokens: the reference embeds all tokens and throws 75 % away (``cinema/mae/mae.py:548-550``); the result for
the kept ones is identical.  ``MultiScaleFusion`` likewise projects only the kept patches of each skip map.
"""

from __future__ import annotations

import math
import os
from pathlib import Path

import torch
import torch.nn.functional as F  # noqa: N812
from torch import nn

from cinema_amd import hip as K
from cinema_amd import tape as T
from cinema_amd.conv import CompactVolume, Conv2d, Conv3d, ConvNormActBlock, Linear, MaskedConvBlock, Volume, _CkptFlag
from cinema_amd.vit import PatchEmbed, get_pos_embed, init_weights

def upsample_mask(mask: torch.Tensor, scale_factor: tuple) -> torch.Tensor:
    """Nearest-neighbour upsampling of a (batch, *grid) bool mask (reference ``cinema/convvit.py:24-51``)."""
    if mask.ndim != len(scale_factor) + 1:
        raise ValueError(f"mask must have the same number of dimensions as scale_factor except batch, got {mask.ndim} and {len(scale_factor)}.")
    for axis, f in enumerate(scale_factor):
        mask = mask.repeat_interleave(int(f), dim=axis + 1)
    return mask


DENSE_STEM = bool(int(os.environ.get("CINEMA_DENSE_STEM", "0")))


def _raster(u: tuple, dims: tuple) -> int:
    r = 0
    for a, d in zip(u, dims):
        r = r * d + a
    return r


def hierarchical_positions(patch_sizes: list, level: int) -> list:
    """Row offset inside a token's block for every stage-``level`` voxel (raster order over the block): voxels are nested
    coarse -> fine so that the children of a stage-(level+1) voxel are contiguous, in raster order (then a k == s conv over
    them is a plain reshape of the compact rows)."""
    n = len(patch_sizes) - 1
    block = tuple(math.prod(ps[d] for ps in patch_sizes[level:]) for d in range(len(patch_sizes[0])))

    def off(lv: int, u: tuple) -> int:
        if lv == n:
            return _raster(u, patch_sizes[n])
        f = patch_sizes[lv]
        return off(lv + 1, tuple(a // b for a, b in zip(u, f))) * math.prod(f) + _raster(tuple(a % b for a, b in zip(u, f)), f)

    coords = [()]
    for d in block:
        coords = [c + (i,) for c in coords for i in range(d)]
    return [off(level, u) for u in coords]


class TokenSelection:
    """Index bookkeeping for one view's random mask, computed once per forward with integer tensor ops (no host sync).

    ``keep`` / ``drop``: int32 flat token ids ``b * n_patches + i`` in raster order (the order boolean-mask indexing
    yields in the reference, ``cinema/mae/mae.py:550``); ``*_pos``: the same ids modulo n_patches (rows of a pos table).
    """

    def __init__(self, mask: torch.Tensor | None, batch: int, n_patches: int, device: torch.device, n_masked: int | None = None) -> None:
        self.mask, self.batch, self.n_patches = mask, batch, n_patches
        if mask is None:
            self.n_keep, self.n_drop = n_patches, 0
            self.keep_pos = T.const(("sel_pos", batch, n_patches, str(device)), lambda: torch.arange(n_patches, dtype=torch.int32, device=device).repeat(batch))
            self.keep = T.const(("sel_all", batch, n_patches, str(device)), lambda: torch.arange(batch * n_patches, dtype=torch.int32, device=device))
            self.drop = self.drop_pos = T.const(("sel_none", str(device)), lambda: torch.empty(0, dtype=torch.int32, device=device))
            self.all_tokens = True
            return
        # every row of a mask from get_batch_random_patch_mask has the same count; unknown (injected) masks are read back once
        if n_masked is not None:
            self.n_drop = int(n_masked)
        else:  # injected mask: one read-back, which also checks that every sample masks the same number of patches (the reference fails in
            counts = mask.sum(dim=1)  # its reshape at mae.py:550 when they differ; unequal rows would give index lists of the wrong length here)
            self.n_drop = int(counts[0])
            if bool((counts != counts[0]).any()):
                raise ValueError(f"every sample must mask the same number of patches, got per-sample counts {counts.tolist()}")
        self.n_keep = n_keep = n_patches - self.n_drop

        if mask.is_cuda:  # one launch: raster-ordered kept / dropped lists (a recorded launch of a recorded step)
            self.keep_pos, self.drop_pos, self.keep, self.drop = K.mask_select(mask.contiguous(), n_keep)
        else:  # host-logic tests
            base = torch.arange(batch, dtype=torch.int32, device=device)[:, None] * n_patches
            order = torch.argsort(mask.to(torch.uint8), dim=1, stable=True).to(torch.int32)  # kept (0) first, raster order preserved
            self.keep_pos, self.drop_pos = order[:, :n_keep].reshape(-1).contiguous(), order[:, n_keep:].reshape(-1).contiguous()
            self.keep, self.drop = (base + order[:, :n_keep]).reshape(-1).contiguous(), (base + order[:, n_keep:]).reshape(-1).contiguous()
        self.all_tokens = False


class DownsampleEncoder(nn.Module, _CkptFlag):
    """ConvMAE-style stem + patch embedding (reference ``cinema/convvit.py:54-207``)."""

    def __init__(self, image_size: tuple, in_chans: int, patch_size: tuple, scale_factor: tuple, conv_chans: list, conv_n_blocks: int, embed_dim: int,
                 norm: str) -> None:
        super().__init__()
        n_dims = len(image_size)
        self.patch_sizes = [tuple(patch_size)] + [tuple(scale_factor)] * len(conv_chans)
        size, eff, chans = tuple(image_size), (1,) * n_dims, in_chans
        self.conv_blocks = nn.ModuleList()
        for patch_i, chans_i in zip(self.patch_sizes[:-1], conv_chans):
            block = nn.Module()
            block.patch_embed = ConvNormActBlock(n_dims=n_dims, in_chans=chans, out_chans=chans_i, norm=norm, kernel_size=patch_i, stride=patch_i,
                                                 padding="valid")
            size = tuple(s // p for s, p in zip(size, patch_i))
            eff = tuple(s * p for s, p in zip(eff, patch_i))
            chans = chans_i
            block.conv = nn.ModuleList([MaskedConvBlock(n_dims=n_dims, in_chans=chans_i, norm=norm) for _ in range(conv_n_blocks)])
            self.conv_blocks.append(block)
        self.eff_patch_size = tuple(s * p for s, p in zip(eff, self.patch_sizes[-1]))
        self.in_chans = in_chans
        self.patch_embed = PatchEmbed(image_size=size, patch_size=self.patch_sizes[-1], in_chans=chans, embed_dim=embed_dim)
        self.linear = Linear(embed_dim, embed_dim)
        self.pos_embed = get_pos_embed(embed_dim=embed_dim, grid_size=self.patch_embed.grid_size)
        self.apply(init_weights)

    def set_grad_ckpt(self, enable: bool = True) -> None:
        self.grad_ckpt = enable
        for block in self.conv_blocks:
            block.patch_embed.set_grad_ckpt(enable)
            for conv in block.conv:
                conv.set_grad_ckpt(enable)
        self.patch_embed.set_grad_ckpt(enable)
        self.linear.set_grad_ckpt(enable)

    def interpolate_pos_encoding(self, grid_size: tuple) -> torch.Tensor:
        """(1, n, E) table for an input grid that differs from the built one (reference ``convvit.py:140-163``).  The table is
        a frozen constant, so the resampling is host-side preparation (bicubic 2-D / trilinear 3-D, like the reference)."""
        if tuple(grid_size) == tuple(self.patch_embed.grid_size):
            return self.pos_embed
        key = (tuple(grid_size), self.pos_embed.device, self.pos_embed._version)
        cache = self.__dict__.setdefault("_pe_cache", {})
        if key not in cache:  # resampled ONCE per grid on the host (a few hundred KB), then kept on the device: no ATen kernel on the step's path
            mode = {2: "bicubic", 3: "trilinear"}[len(grid_size)]
            emb = self.pos_embed.shape[-1]
            pe = self.pos_embed.detach().float().cpu().reshape(1, *self.patch_embed.grid_size, emb).movedim(-1, 1)
            pe = F.interpolate(pe, size=tuple(grid_size), mode=mode, antialias=False)
            cache.clear()
            host = pe.movedim(1, -1).reshape(1, -1, emb).contiguous()
            cache[key] = K.persistent(lambda: host.to(device=self.pos_embed.device, dtype=self.pos_embed.dtype))  # outside a recording's private pool
        return cache[key]

    def grid_for(self, image_size: tuple) -> tuple:
        return tuple(s // p for s, p in zip(image_size, self.eff_patch_size))

    def tape_forward(self, tp: T.Tape, image: torch.Tensor, sel: TokenSelection, grid: tuple):  # noqa: ANN201
        """-> (skips: list[Volume], tokens: Var fp32 [b*n_keep, E] WITHOUT the positional table, which the caller adds while
        assembling the encoder sequence)."""
        batch, chans, *size = image.shape
        if sel.mask is not None and not sel.all_tokens and not DENSE_STEM and sel.n_keep > 0:
            return self.tape_forward_visible(tp, image, sel, grid)
        vis_masks: list = [None] * len(self.conv_blocks)
        if sel.mask is not None:
            m = sel.mask.reshape(batch, *grid)
            for lvl in range(len(self.patch_sizes) - 1, 0, -1):  # coarse -> fine (convvit.py:186-192)
                m = upsample_mask(m, self.patch_sizes[lvl])
                vis_masks[lvl - 1] = (~m).reshape(-1).to(torch.uint8).contiguous()
        skips = []
        src, src_chans, src_size, src_strides = T.Var(image, needs_grad=False), chans, tuple(size), tuple(image.stride())
        vol = None
        for block, vis in zip(self.conv_blocks, vis_masks):
            vol = block.patch_embed.tape_forward(tp, src, batch, src_chans, src_size, src_strides)
            for conv in block.conv:
                vol = conv.tape_forward(tp, vol, vis)
            skips.append(vol)
            src, src_chans, src_size, src_strides = vol.var, vol.chans, vol.spatial, vol.strides()
        self.patch_embed.check_size(src_size)
        geom = K.patch_geom(batch, src_chans, grid, self.patch_sizes[-1], src_strides, token_idx=None if sel.all_tokens else sel.keep)
        rows = T.op_patch_gather(tp, src, geom)
        tok = T.op_linear(tp, rows, self.patch_embed.proj.weight, self.patch_embed.proj.bias)
        tok = T.op_linear(tp, tok, self.linear.weight, self.linear.bias, out_f32=True)
        return skips, tok

    def _stage_tables(self, device: torch.device) -> list:
        """Per stage: (block, pos, inv_pos) - voxels per token and the hierarchical row-offset table (constant per model)."""
        cache = getattr(self, "_stage_tables_cache", None)
        if cache is not None and cache[0] == device:
            return cache[1]
        n = len(self.conv_blocks)
        n_dims = len(self.patch_sizes[0])
        tables = []
        for lvl in range(1, n + 1):
            block = tuple(math.prod(ps[d] for ps in self.patch_sizes[lvl:]) for d in range(n_dims))
            pos = hierarchical_positions(self.patch_sizes, lvl)
            inv = [0] * len(pos)
            for u, q in enumerate(pos):
                inv[q] = u
            tables.append((block, torch.tensor(pos, dtype=torch.int32, device=device), torch.tensor(inv, dtype=torch.int32, device=device)))
        self._stage_tables_cache = (device, tables)
        return tables

    def tape_forward_visible(self, tp: T.Tape, image: torch.Tensor, sel: TokenSelection, grid: tuple):  # noqa: ANN201
        """The stem on the visible voxels (see the module docstring): -> (skips: list[CompactVolume], tokens [b*n_keep, E])."""
        batch, chans, *size = image.shape
        dev = image.device
        n_dims = len(size)
        n_tok_all = math.prod(grid)
        n_tok = sel.keep.numel()
        tables = self._stage_tables(dev)
        block1, pos1, inv1 = tables[0]
        grid1 = tuple(g * b for g, b in zip(grid, block1))

        # compact rank of every token (-1: masked) and the stage-1 voxels of the kept tokens, in compact row order, as flat ids of the
        # (batch, *grid1) stage-1 volume: one launch
        rank, idx1 = K.visible_index(sel.keep, batch, grid, block1, inv1)

        # the stages' sparse geometries up front: their neighbour lists depend on the mask only and are built beside the gather / patch GEMM / LayerNorm chain
        geoms = [K.sparse_geom(batch, grid, blk, sel.keep, rank, pos) for blk, pos, _ in tables]
        # neighbour lists of the visible-voxel depthwise convolutions are built on the long-axis stream, beside the chain that precedes their first use
        if T.LAX_STREAM and K.LANE is None and image.is_cuda and not torch._C._cuda_isCurrentStreamCapturing():  # (a lane group's launches go out later, zipped)
            items, seen = [], set()
            for block, sg in zip(self.conv_blocks, geoms):
                for conv in block.conv:
                    kd = (1,) * (3 - n_dims) + tuple(int(v) for v in conv.dw_conv.weight.shape[2:])
                    if K.sparse_pair_form(sg, conv.dw_conv.weight.shape[0], kd):  # token-pair kernels (csrc/stem_dw.hip): no list to build
                        continue
                    if (id(sg), kd) not in seen:
                        seen.add((id(sg), kd))
                        items.append((sg, kd))
            if items:
                K.sparse_nbr_prefetch(items, dev, T.lax_stream().cuda_stream)
        skips = []
        vol = None
        for lvl, (block, (blk, pos, inv)) in enumerate(zip(self.conv_blocks, tables)):
            if lvl == 0:
                geom = K.patch_geom(batch, chans, grid1, self.patch_sizes[0], tuple(image.stride()), token_idx=idx1)
                rows = T.op_patch_gather(tp, T.Var(image, needs_grad=False), geom)
            else:
                per = math.prod(self.patch_sizes[lvl])
                rows = T.op_cast_bf16(tp, T.op_view(tp, vol.var, (vol.var.data.shape[0] // per, per * vol.chans)))
            out = block.patch_embed.tape_forward_rows(tp, rows)
            sg = geoms[lvl]
            vol = CompactVolume(out, n_tok, blk, block.patch_embed.conv.out_channels, sg, pos, inv)
            for conv in block.conv:
                vol = conv.tape_forward_compact(tp, vol)
            skips.append(vol)
        tok = T.op_linear(tp, vol.token_rows(tp), self.patch_embed.proj.weight, self.patch_embed.proj.bias)
        tok = T.op_linear(tp, tok, self.linear.weight, self.linear.bias, out_f32=True)
        return skips, tok

    def forward(self, image: torch.Tensor, mask: torch.Tensor | None):  # noqa: ANN201
        """Reference signature: returns (skips as channels-first tensors, tokens (batch, n_patches, E) for ALL tokens)."""
        batch = image.shape[0]
        grid = self.grid_for(tuple(image.shape[2:]))
        n = math.prod(grid)
        dev = image.device

        def run(tp: T.Tape):  # noqa: ANN202
            full = TokenSelection(None, batch, n, dev)
            full.mask = mask  # stem masking still applies; every token is embedded
            skips, tok = self.tape_forward(tp, image.float().contiguous(), full, grid)
            pe = self.interpolate_pos_encoding(grid).detach().reshape(n, -1)
            out = T.op_assemble(tp, batch * n, tok.data.shape[1], [T.Segment(full.keep, src=tok, add=pe, add_idx=full.keep_pos)], dev)
            return [s.var for s in skips] + [out], []

        outs = T.taped_call(run, [], list(self.parameters()))
        sk, tok = outs[:-1], outs[-1]
        size = tuple(image.shape[2:])
        skips = []
        for lvl, s in enumerate(sk):
            size = tuple(v // p for v, p in zip(size, self.patch_sizes[lvl]))
            skips.append(s.reshape(batch, *size, -1).movedim(-1, 1).contiguous())
        return skips, tok.reshape(batch, n, -1)


class MultiScaleFusion(nn.Module, _CkptFlag):
    """x + sum_i down_i(skip_i)[kept] -> LayerNorm (reference ``cinema/convvit.py:210-291``)."""

    def __init__(self, image_size: tuple, patch_size: tuple, scale_factor: tuple, conv_chans: list, embed_dim: int, norm_layer: type,
                 norm_eps: float) -> None:
        super().__init__()
        n_dims = len(image_size)
        patch_sizes = [tuple(patch_size)] + [tuple(scale_factor)] * len(conv_chans)
        grid = tuple(image_size)
        for p in patch_sizes:
            grid = tuple(s // q for s, q in zip(grid, p))
        size = tuple(image_size)
        conv_cls = Conv2d if n_dims == 2 else Conv3d
        self.down_convs = nn.ModuleList()
        for i, ch in enumerate(conv_chans):
            size = tuple(s // p for s, p in zip(size, patch_sizes[i]))
            kernel = tuple(s // g for s, g in zip(size, grid))
            self.down_convs.append(conv_cls(ch, embed_dim, kernel_size=kernel, stride=kernel, padding="valid"))
        self.norm = norm_layer(embed_dim, eps=norm_eps)
        self.apply(init_weights)

    def set_grad_ckpt(self, enable: bool = True) -> None:
        self.grad_ckpt = enable
        for conv in self.down_convs:
            conv.set_grad_ckpt(enable)

    def tape_forward(self, tp: T.Tape, skips: list, x: T.Var, sel: TokenSelection, grid: tuple, out_f32: bool = False) -> T.Var:
        """x: fp32 [b*n_keep, E] rows of this view (after ``encoder.norm``).  Output bf16 (decoder path) or fp32 (features)."""
        for vol, conv in zip(skips, self.down_convs):
            if isinstance(vol, CompactVolume):  # visible-voxel stem: one row per kept token already, voxels in hierarchical order
                x = T.op_linear(tp, vol.token_rows(tp), conv.weight, conv.bias, residual=x, w16=T.w_patch_perm(conv.weight, vol.inv_pos),
                                to_param_layout=T.patch_grad_to_param_perm(conv.weight, vol.pos, vol.inv_pos))
                continue
            geom = K.patch_geom(vol.batch, vol.chans, grid, tuple(conv.kernel_size), vol.strides(), token_idx=None if sel.all_tokens else sel.keep)
            rows = T.op_patch_gather(tp, vol.var, geom)
            x = T.op_linear(tp, rows, conv.weight, conv.bias, residual=x, w16=T.w_patch(conv.weight), to_param_layout=T.patch_grad_to_param(conv.weight))
        return T.op_layernorm(tp, x, self.norm.weight, self.norm.bias, self.norm.eps, out_f32=out_f32)


def stem_geometry(model, v: str, images: dict, sels: dict) -> tuple:  # noqa: ANN001
    """Everything that determines the launch sequence of a view's stem / fusion / head ops: views with equal keys issue identical sequences."""
    enc = model.enc_down_dict[v]
    shapes = enc.__dict__.get("_cinema_param_shapes")
    if shapes is None:
        shapes = enc.__dict__["_cinema_param_shapes"] = tuple(tuple(p.shape) for p in enc.parameters())
    s = sels[v]
    return (tuple(images[v].shape), tuple(enc.patch_sizes), shapes, s.n_keep, s.n_drop, s.all_tokens, s.mask is None)


def encode_views(model, tp: T.Tape, views: list, images: dict, sels: dict, grids: dict):  # noqa: ANN001, ANN201
    """Stem -> kept-token embedding (+ positional table) -> [cls | view tokens] sequence -> ``model.encoder`` -> LN, shared by
    ``CineMA`` and ``ConvViT`` (reference ``mae.py:535-562``, ``convvit.py:478-493``).
    Returns (x fp32 [b*T, E], skips per view, cls row indices, row indices per view)."""
    batch = sels[views[0]].batch
    dev = images[views[0]].device
    e = model.encoder.cls_token.shape[-1]
    n_keep = [sels[v].n_keep for v in views]
    t_e = 1 + sum(n_keep)

    def rows_of(off: int, n: int) -> torch.Tensor:  # row b * t_e + off + i for i < n: shape-only, cached
        return T.const(("rows_of", batch, t_e, off, n, str(dev)), lambda: (
            torch.arange(batch, dtype=torch.int32, device=dev)[:, None] * t_e + off + torch.arange(n, dtype=torch.int32, device=dev)[None]).reshape(-1).contiguous())

    cls_rows = rows_of(0, 1)
    segs, skips_all, view_rows = [T.Segment(cls_rows, src=model.encoder.cls_token)], {}, {}
    offs, off = {}, 1
    for v, nk in zip(views, n_keep):
        offs[v] = off
        off += nk

    def stem(v: str) -> None:
        enc = model.enc_down_dict[v]
        skips_all[v], tok = enc.tape_forward(tp, images[v], sels[v], grids[v])
        rows = rows_of(offs[v], sels[v].n_keep)
        view_rows[v] = rows
        pe = enc.interpolate_pos_encoding(grids[v]).detach().reshape(-1, e)
        segs.append(T.Segment(rows, src=tok, add=pe, add_idx=sels[v].keep_pos))

    # consecutive views of identical geometry (the three long-axis views: same shapes, separate weights) run as ONE lane group: their ~45 forward /
    # ~95 backward tiny stem launches each go out zipped, one wide launch per position (hip.lanes; the views share nothing inside the stems)
    # ... and that group goes to a stream of its own, beside the short-axis stem's chain (tape.LAX_STREAM)
    T.run_in_lanes(tp, list(views), lambda v: stem_geometry(model, v, images, sels), stem, enabled=images[views[0]].is_cuda, beside=True)
    x = T.op_assemble(tp, batch * t_e, e, segs, dev)
    x = model.encoder.tape_forward(tp, x, batch)
    return x, skips_all, cls_rows, view_rows


# --------------------------------------------------------------------------------------------------------------
# ConvViT: multi-view classifier / regressor on the same stem + encoder + fusion (reference cinema/convvit.py:294-810)
# --------------------------------------------------------------------------------------------------------------
def get_model(config) -> "ConvViT":  # noqa: ANN001
    """Same config mapping as the reference ``get_model`` (``convvit.py:294-334``); ``config`` needs attribute access."""
    from cinema_amd.vit import get_vit_config

    views = [config.model.views] if isinstance(config.model.views, str) else list(config.model.views)
    vit = get_vit_config(config.model.convvit.size)
    in_chans_dict = {v: config.data.sax.in_chans if v == "sax" else config.data.lax.in_chans for v in views}
    if hasattr(config.data, "class_column"):
        out_chans = len(config.data[config.data.class_column])
    elif hasattr(config.data, "regression_column"):
        out_chans = 1
    else:
        out_chans = config.model.out_chans
    image_size_dict = {v: tuple(config.data.sax.patch_size if v == "sax" else config.data.lax.patch_size) for v in views}
    ndim = {v: 3 if v == "sax" else 2 for v in views}
    model = ConvViT(image_size_dict=image_size_dict, n_frames=config.model.n_frames, in_chans_dict=in_chans_dict, out_chans=out_chans,
                    enc_patch_size_dict={v: tuple(config.model.convvit.enc_patch_size[:n]) for v, n in ndim.items()},
                    enc_scale_factor_dict={v: tuple(config.model.convvit.enc_scale_factor[:n]) for v, n in ndim.items()},
                    enc_conv_chans=list(config.model.convvit.enc_conv_chans), enc_conv_n_blocks=config.model.convvit.enc_conv_n_blocks,
                    enc_embed_dim=vit["enc_embed_dim"], enc_depth=vit["enc_depth"], enc_n_heads=vit["enc_n_heads"],
                    drop_path=config.model.convvit.drop_path)
    model.set_grad_ckpt(config.grad_ckpt)
    return model


class ConvViT(nn.Module):
    """Multi-view ViT with the ConvMAE stem for classification / regression (reference ``cinema/convvit.py:337-614``)."""

    def __init__(self, image_size_dict: dict, in_chans_dict: dict, n_frames: int, out_chans: int, enc_patch_size_dict: dict,
                 enc_scale_factor_dict: dict, enc_conv_chans: list, enc_conv_n_blocks: int, enc_embed_dim: int, enc_depth: int, enc_n_heads: int,
                 mlp_ratio: int = 4, qkv_bias: bool = True, norm_layer: type = nn.LayerNorm, norm_eps: float = 1e-5, rotary: bool = False,
                 act_layer: type = nn.GELU, mlp_layer: type | None = None, drop_path: float = 0.0, norm: str = "layer",
                 head_layer: type | None = nn.Linear) -> None:
        from cinema_amd.vit import Mlp, ViTEncoder

        super().__init__()
        self.grad_ckpt = False
        self.views = list(image_size_dict.keys())
        self.n_frames = n_frames
        self.enc_down_dict = nn.ModuleDict({
            v: DownsampleEncoder(image_size=tuple(image_size_dict[v]), in_chans=n_frames * in_chans_dict[v], patch_size=tuple(enc_patch_size_dict[v]),
                                 scale_factor=tuple(enc_scale_factor_dict[v]), conv_chans=enc_conv_chans, conv_n_blocks=enc_conv_n_blocks,
                                 embed_dim=enc_embed_dim, norm=norm) for v in self.views})
        self.enc_fusion_dict = nn.ModuleDict({
            v: MultiScaleFusion(image_size=tuple(image_size_dict[v]), patch_size=tuple(enc_patch_size_dict[v]),
                                scale_factor=tuple(enc_scale_factor_dict[v]), conv_chans=enc_conv_chans, embed_dim=enc_embed_dim,
                                norm_layer=norm_layer, norm_eps=norm_eps) for v in self.views})
        self.encoder = ViTEncoder(embed_dim=enc_embed_dim, depth=enc_depth, n_heads=enc_n_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                  norm_layer=norm_layer, norm_eps=norm_eps, rotary=rotary, act_layer=act_layer, mlp_layer=mlp_layer or Mlp,
                                  drop_path=drop_path)
        self.apply(init_weights)
        # heads are created AFTER apply(init_weights): they keep torch's default Linear init (reference convvit.py:439-445)
        self.pred_head_dict = nn.ModuleDict()
        if head_layer is not None:
            for v in [*self.views, "cls"]:
                self.pred_head_dict[v] = head_layer(enc_embed_dim, out_chans)

    def set_grad_ckpt(self, enable: bool = True) -> None:
        """Accepted for API compatibility (``convvit.py:447-457``); nothing is recomputed on this path."""
        self.grad_ckpt = enable
        for v in self.views:
            self.enc_down_dict[v].set_grad_ckpt(enable)
            self.enc_fusion_dict[v].set_grad_ckpt(enable)
        self.encoder.set_grad_ckpt(enable)

    def _features(self, tp: T.Tape, views: list, images: dict, mask_dict: dict | None):  # noqa: ANN202
        """-> (cls rows Var [b, E], {view: Var [b*n_view, E]}, n tokens per view): all tokens are embedded, ``mask_dict`` only masks
        the conv stem; the fusion sees every token (``mask=None``, reference ``convvit.py:501``)."""
        batch = images[views[0]].shape[0]
        dev = images[views[0]].device
        grids = {v: self.enc_down_dict[v].grid_for(tuple(images[v].shape[2:])) for v in views}
        sels = {}
        for v in views:
            sels[v] = TokenSelection(None, batch, math.prod(grids[v]), dev)
            if mask_dict is not None:
                sels[v].mask = mask_dict[v].to(device=dev, dtype=torch.bool)
        x, skips_all, cls_rows, view_rows = encode_views(self, tp, views, images, sels, grids)
        parts = T.op_split_rows(tp, x, [cls_rows] + [view_rows[v] for v in views])
        feats = {}
        for i, v in enumerate(views):
            feats[v] = self.enc_fusion_dict[v].tape_forward(tp, skips_all[v], parts[i + 1], sels[v], grids[v], out_f32=True)
        return parts[0], feats, {v: sels[v].n_keep for v in views}

    def _check(self, image_dict: dict) -> list:
        views = list(image_dict.keys())
        if any(v not in self.views for v in views):
            raise ValueError(f"views {views} must be in self.input_keys {self.views}.")
        return views

    def feature_forward(self, image_dict: dict, mask_dict: dict | None) -> dict:
        """{"cls": (b, 1, E), view: (b, n_patches_view, E)} (reference ``convvit.py:459-503``)."""
        views = self._check(image_dict)
        batch = image_dict[views[0]].shape[0]
        images = {v: image_dict[v].float().contiguous() for v in views}

        def run(tp: T.Tape):  # noqa: ANN202
            cls, feats, _ = self._features(tp, views, images, mask_dict)
            return [cls] + [feats[v] for v in views], []

        res = T.taped_call(run, [], T.trainable_params(self))
        return {k: r.reshape(batch, -1, r.shape[-1]) for k, r in zip(["cls", *views], res)}

    def forward(self, image_dict: dict, mask_dict: dict | None = None, reduce: str = "all") -> torch.Tensor:
        """logits (batch, out_chans); ``reduce`` in {"patch", "all", "cls"} (reference ``convvit.py:505-558``)."""
        if reduce not in {"patch", "all", "cls"}:
            raise NotImplementedError(f"Unsupported reduce method {reduce}.")
        views = self._check(image_dict)
        if reduce != "cls" and any(v not in image_dict for v in self.views):
            raise KeyError(f"reduce='{reduce}' averages the heads of all model views {self.views}")  # the reference indexes x_dict[view] for every view
        batch = image_dict[views[0]].shape[0]
        images = {v: image_dict[v].float().contiguous() for v in views}

        def run(tp: T.Tape):  # noqa: ANN202
            T.begin_stochastic(self, images[views[0]].device)  # drop-path of the fine-tuning recipes
            cls, feats, _ = self._features(tp, views, images, mask_dict)
            if reduce == "cls":
                head = self.pred_head_dict["cls"]
                return [T.op_linear(tp, T.op_cast_bf16(tp, cls), head.weight, head.bias, out_f32=True)], []
            acc = None
            for v in self.views:  # mean over tokens, head, then mean over the heads
                head = self.pred_head_dict[v]
                pooled = T.op_cast_bf16(tp, T.op_segment_mean(tp, feats[v], batch))
                acc = T.op_linear(tp, pooled, head.weight, head.bias, out_f32=True, residual=acc)
            k = len(self.views)
            if reduce == "all":
                head = self.pred_head_dict["cls"]
                acc = T.op_linear(tp, T.op_cast_bf16(tp, cls), head.weight, head.bias, out_f32=True, residual=acc)
                k += 1
            return [T.op_scale(tp, acc, 1.0 / k)], []

        (logits,) = T.taped_call(run, [], T.trainable_params(self))
        return logits.reshape(batch, -1)

    @classmethod
    def from_finetuned(cls, repo_id: str | None = None, model_filename: str | None = None, config_filename: str | None = None, *,
                       model_path: str | Path | None = None, config_path: str | Path | None = None, **kwargs) -> "ConvViT":  # noqa: ANN003
        """Fine-tuned weights + config (reference ``convvit.py:560-593``); pass local ``model_path`` / ``config_path`` on an air-gapped box."""
        import yaml
        from safetensors.torch import load_file

        from cinema_amd.config import to_config

        if model_path is None or config_path is None:
            from huggingface_hub import hf_hub_download

            model_path = model_path or hf_hub_download(repo_id=repo_id, filename=model_filename, **kwargs)
            config_path = config_path or hf_hub_download(repo_id=repo_id, filename=config_filename, **kwargs)
        with open(config_path, encoding="utf-8") as f:
            config = to_config(yaml.safe_load(f))
        model = get_model(config)
        model.load_state_dict(load_file(str(model_path)))
        return model

    @classmethod
    def from_pretrained(cls, config, freeze: bool, model_path: str | Path | None = None, **kwargs) -> "ConvViT":  # noqa: ANN001, ANN003
        """MAE-pretrained stem / encoder / fusion weights into a fresh classifier (reference ``convvit.py:595-613``)."""
        if model_path is None:
            from huggingface_hub import hf_hub_download

            model_path = hf_hub_download(repo_id="mathpluscode/CineMA", filename="pretrained/cinema.safetensors", **kwargs)
        return load_pretrain_weights(model=get_model(config), views=config.model.views, ckpt_path=Path(model_path), freeze=freeze)


def load_pretrain_weights(model: nn.Module, views, ckpt_path: Path, freeze: bool) -> nn.Module:  # noqa: ANN001
    """Load MAE weights into a downstream model (reference ``convvit.py:616-704``): decoder / heads / other views / positional tables
    are dropped, the first stem conv is tiled over extra input channels, the only keys allowed to be missing are the views' ``pos_embed``."""
    ckpt_path = Path(ckpt_path)
    if ckpt_path.suffix == ".pt":
        pretrained = torch.load(ckpt_path, map_location="cpu")["model"]
    elif ckpt_path.suffix == ".safetensors":
        from safetensors.torch import load_file

        pretrained = load_file(str(ckpt_path))
    else:
        raise ValueError(f"Unsupported checkpoint type {ckpt_path.suffix}.")
    keys_to_drop = ["mask", "decoder", "_head", "sax", "lax_2c", "lax_3c", "lax_4c", "fusion", "dec_linear", "pos_embed"]
    if hasattr(model, "enc_fusion_dict"):
        keys_to_drop.remove("fusion")
    views = [views] if isinstance(views, str) else list(views)
    expected_missing = []
    for v in views:
        keys_to_drop.remove(v)
        expected_missing.append(f"enc_down_dict.{v}.pos_embed")
    state = {}
    for k, val in pretrained.items():
        if any(x in k for x in keys_to_drop):
            continue
        for v in views:
            if k == f"enc_down_dict.{v}.conv_blocks.0.patch_embed.conv.weight":
                chans = model.enc_down_dict[v].conv_blocks[0].patch_embed.conv.weight.shape[1]
                if val.shape[1] != chans:  # video / multi-modality input: tile the single-channel filter (convvit.py:663-681)
                    if val.dim() not in (4, 5):
                        raise ValueError(f"Unsupported weight shape {val.shape}.")
                    val = val.repeat(1, chans, *([1] * (val.dim() - 2)))
        state[k] = val
    incompatible = model.load_state_dict(state, strict=False)
    missing = [x for x in incompatible.missing_keys if ("decoder" not in x) and (not x.startswith("dec_")) and ("head" not in x)]
    if set(missing) != set(expected_missing):
        raise ValueError(f"Missing keys from checkpoint: {missing}, expected {expected_missing}")
    if len(incompatible.unexpected_keys) > 0:
        raise ValueError(f"Unexpected keys in checkpoint: {incompatible.unexpected_keys}")
    if freeze:
        for name, param in model.named_parameters():
            if name in state:
                param.requires_grad = False
    return model


def get_layer_id_for_vit(name: str, n_layers: int) -> int:
    """Layer id for layer-wise lr decay, first layer is 1 (reference ``convvit.py:707-738``)."""
    if name.startswith("enc_"):
        return 0
    if any(x in name for x in ["cls_token", "pos_embed", "patch_embed", "view_embed"]):
        return 0
    if name.startswith("encoder.blocks"):
        return int(name.split(".")[2]) + 1
    return n_layers


def param_groups_lr_decay(model: nn.Module, no_weight_decay_list: list, weight_decay: float, layer_decay: float, out_dir: Path | None = None) -> list:
    """Parameter groups with layer-wise lr decay (reference ``convvit.py:741-810``): ``lr_scale = layer_decay ** (n_layers - layer_id)``."""
    import json

    names: dict = {}
    groups: dict = {}
    n_layers = len(model.encoder.blocks) + 1
    scales = [layer_decay ** (n_layers - i) for i in range(n_layers + 1)]
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        decay_name, this_decay = ("no_decay", 0.0) if (p.ndim == 1 or n in no_weight_decay_list) else ("decay", weight_decay)
        layer_id = get_layer_id_for_vit(n, n_layers)
        key = f"layer_{layer_id}_{decay_name}"
        if key not in groups:
            names[key] = {"lr_scale": scales[layer_id], "weight_decay": this_decay, "params": []}
            groups[key] = {"lr_scale": scales[layer_id], "weight_decay": this_decay, "params": []}
        names[key]["params"].append(n)
        groups[key]["params"].append(p)
    if out_dir is not None:
        out_dir = Path(out_dir)
        out_dir.mkdir(parents=True, exist_ok=True)
        with open(out_dir / "param_group_names.json", "w", encoding="utf-8") as f:
            json.dump(names, f, indent=2)
    return list(groups.values())
