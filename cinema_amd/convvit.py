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
        base = torch.arange(batch, dtype=torch.int32, device=device)[:, None] * n_patches
        if mask is None:
            pos = torch.arange(n_patches, dtype=torch.int32, device=device)[None].expand(batch, -1)
            self.n_keep, self.n_drop = n_patches, 0
            self.keep_pos = pos.reshape(-1).contiguous()
            self.keep = (base + pos).reshape(-1).contiguous()
            self.drop = self.drop_pos = torch.empty(0, dtype=torch.int32, device=device)
            self.all_tokens = True
            return
        order = torch.argsort(mask.to(torch.uint8), dim=1, stable=True).to(torch.int32)  # kept (0) first, raster order preserved
        # every row of a mask from get_batch_random_patch_mask has the same count; unknown (injected) masks are read back once
        self.n_drop = int(n_masked) if n_masked is not None else int(mask[0].sum())
        self.n_keep = n_patches - self.n_drop
        self.keep_pos = order[:, :self.n_keep].reshape(-1).contiguous()
        self.drop_pos = order[:, self.n_keep:].reshape(-1).contiguous()
        self.keep = (base + order[:, :self.n_keep]).reshape(-1).contiguous()
        self.drop = (base + order[:, self.n_keep:]).reshape(-1).contiguous()
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
        mode = {2: "bicubic", 3: "trilinear"}[len(grid_size)]
        emb = self.pos_embed.shape[-1]
        pe = self.pos_embed.float().reshape(1, *self.patch_embed.grid_size, emb).movedim(-1, 1)
        pe = F.interpolate(pe, size=tuple(grid_size), mode=mode, antialias=False)
        return pe.movedim(1, -1).reshape(1, -1, emb).to(self.pos_embed.dtype)

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
        rank = torch.full((batch * n_tok_all,), -1, dtype=torch.int32, device=dev)
        rank[sel.keep.long()] = torch.arange(n_tok, dtype=torch.int32, device=dev)
        # stage-1 voxels of the kept tokens, in compact row order, as flat ids of the (batch, *grid1) stage-1 volume
        block1, pos1, inv1 = tables[0]
        grid1 = tuple(g * b for g, b in zip(grid, block1))
        t = sel.keep.long() % n_tok_all
        bb = sel.keep.long() // n_tok_all
        tcoord = []
        for g in reversed(grid):
            tcoord.append(t % g)
            t = t // g
        tcoord.reverse()
        u = inv1.long()  # raster voxel index stored at row offset q
        ucoord = []
        for bdim in reversed(block1):
            ucoord.append(u % bdim)
            u = u // bdim
        ucoord.reverse()
        vid = bb[:, None]
        for d in range(n_dims):
            vid = vid * grid1[d] + (tcoord[d][:, None] * block1[d] + ucoord[d][None, :])
        idx1 = vid.reshape(-1).to(torch.int32).contiguous()

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
            sg = K.sparse_geom(batch, grid, blk, sel.keep, rank, pos)
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
                                to_param_layout=T.patch_grad_to_param_perm(conv.weight, vol.pos))
                continue
            geom = K.patch_geom(vol.batch, vol.chans, grid, tuple(conv.kernel_size), vol.strides(), token_idx=None if sel.all_tokens else sel.keep)
            rows = T.op_patch_gather(tp, vol.var, geom)
            x = T.op_linear(tp, rows, conv.weight, conv.bias, residual=x, w16=T.w_patch(conv.weight), to_param_layout=T.patch_grad_to_param(conv.weight))
        return T.op_layernorm(tp, x, self.norm.weight, self.norm.bias, self.norm.eps, out_f32=out_f32)
