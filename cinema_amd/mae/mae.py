"""CineMA masked autoencoder on the MI355X HIP path (interface of the reference ``cinema/mae/mae.py``).

``CineMA.forward(image_dict, enc_mask_ratio)`` returns ``(loss, pred_dict, enc_mask_dict, metrics)`` exactly like the
reference (``cinema/mae/mae.py:504-612``) and the parameters / ``state_dict`` keys are identical, but the whole forward is
one kernel sequence on :mod:`cinema_amd.tape` and one autograd node:

  per view : conv stem (patch-gather + MFMA GEMM + fused LN/GELU, masked ConvMAE blocks with the depthwise kernel)
             -> kept-token patch embedding (gather + 2 GEMMs)                                  [A1-A7 in SURVEY.md 2.3]
  shared   : assemble [cls | kept tokens + pos] -> 12 encoder blocks -> LN                     [A8-A12]
  per view : multi-scale fusion on kept patches + LN -> dec_linear                             [A13-A14]
  shared   : decoder embedding (mask tokens + pos) -> 8 cross-attention blocks -> LN           [A15-A16]
  per view : prediction head GEMM -> masked-patch MSE with on-the-fly target gather            [A17-A18]

No device->host synchronisation happens inside ``forward`` (the reference has ~17: boolean-mask indexing, per-view
``isfinite`` branches): token selection uses stable-argsort index tensors and the finite-mean of the view losses is a kernel.
"""

from __future__ import annotations

import math
from pathlib import Path

import torch
from torch import nn

from cinema_amd import hip as K
from cinema_amd import tape as T
from cinema_amd.conv import Linear
from cinema_amd.convvit import DownsampleEncoder, MultiScaleFusion, TokenSelection, encode_views, stem_geometry
from cinema_amd.vit import Mlp, ViTDecoder, ViTEncoder, get_pos_embed, get_tokens, get_vit_config, init_weights, patchify


def get_batch_random_patch_mask(batch_size: int, n_patches: int, mask_ratio: float, device: torch.device) -> torch.Tensor:
    """Per-sample random mask, True = removed; exactly ``n - int(n * (1 - ratio))`` True per row (reference ``mae.py:30-65``).

    Same recipe (uniform noise -> rank) and the same RNG stream consumption (one ``torch.rand(batch, n)``) as the reference.
    """
    if mask_ratio < 0:
        raise ValueError(f"mask_ratio must be positive, got {mask_ratio}.")
    if mask_ratio == 0:
        return torch.zeros((batch_size, n_patches), dtype=torch.bool, device=device)
    noise = torch.rand(batch_size, n_patches, device=device)
    n_keep = int(n_patches * (1 - mask_ratio))
    if noise.is_cuda:  # one launch: rank of every element in its row (ties by index) >= n_keep
        return K.random_mask(noise, n_keep)
    rank = torch.argsort(torch.argsort(noise, dim=1, stable=True), dim=1, stable=True)
    return rank >= n_keep


def add_pos_embed_and_append_mask_token(x_vis: torch.Tensor, enc_mask: torch.Tensor, dec_pos_embed: nn.Parameter, mask_token: nn.Parameter, concat: bool):  # noqa: ANN201
    """Decoder input of one view (reference ``cinema/mae/mae.py:68-104``): visible tokens + their rows of the positional table, mask tokens + the rows of the
    masked patches, both in raster order.  ``x_vis`` (batch, n_keep, d), ``enc_mask`` (batch, n_patches) bool with True = masked, ``dec_pos_embed`` (n_patches, d) or
    (1, n_patches, d), ``mask_token`` (1, 1, d).  Returns the concatenation (batch, n_patches, d) or the pair (visible, masked).  One multi-segment row-copy
    launch (``tape.op_assemble``); gradients flow to ``x_vis`` and ``mask_token`` like in the reference."""
    batch, n_keep, d = x_vis.shape
    n_patches = enc_mask.shape[1]
    n_masked = n_patches - n_keep
    dev = x_vis.device
    sel = TokenSelection(enc_mask, batch, n_patches, dev, n_masked=n_masked)
    table = dec_pos_embed.detach().reshape(-1, d)
    per = torch.arange(batch, dtype=torch.int32, device=dev)[:, None] * n_patches
    vis_rows = (per + torch.arange(n_keep, dtype=torch.int32, device=dev)[None]).reshape(-1).contiguous()
    mask_rows = (per + n_keep + torch.arange(n_masked, dtype=torch.int32, device=dev)[None]).reshape(-1).contiguous()

    def run(tp: T.Tape, xv: T.Var):  # noqa: ANN202
        segs = [T.Segment(vis_rows, src=xv, add=table, add_idx=sel.keep_pos)]
        if n_masked > 0:
            segs.append(T.Segment(mask_rows, src=mask_token, add=table, add_idx=sel.drop_pos))
        return [T.op_assemble(tp, batch * n_patches, d, segs, dev)], []

    (out,) = T.taped_call(run, [x_vis.float().reshape(batch * n_keep, d).contiguous()], [mask_token])
    out = out.reshape(batch, n_patches, d)
    if concat:
        return out
    return out[:, :n_keep], out[:, n_keep:]


def mse_loss(target: torch.Tensor, pred: torch.Tensor, enc_mask: torch.Tensor, norm_target: bool, epsilon: float = 1.0e-6) -> tuple:
    """Masked-patch MSE and its metrics (reference ``cinema/mae/mae.py:107-152``): ``target`` (batch, n_patches, f) patches, ``pred`` (batch, n_masked, f),
    ``enc_mask`` (batch, n_patches) bool, True = predicted.  The per-patch statistics (unbiased variance), the optional target normalisation, the squared error and
    its gradient with respect to ``pred`` are the kernels the model's forward uses (``cinema_mse_fwd / _bwd``, ``cinema_patch_stats``): the target is addressed as
    an image whose patches are its rows.  The gradient with respect to ``pred`` comes back rounded to bf16 (what the prediction head's backward GEMMs read)."""
    batch, n_patches, f = target.shape
    dev = target.device
    tgt = target.detach().float().contiguous()
    geom_all = K.patch_geom(batch, 1, (n_patches,), (f,), (n_patches * f, n_patches * f, 1))
    stats = T.zeros(2, torch.float32, dev)
    K.patch_stats(tgt, geom_all, stats)
    metrics = {"target_mean": stats[0], "target_std": stats[1]}
    n_masked = pred.shape[1]
    if n_masked == 0:
        nan = torch.full((), float("nan"), device=dev)
        metrics["mse_loss"] = nan
        return nan, metrics
    sel = TokenSelection(enc_mask, batch, n_patches, dev, n_masked=n_masked)
    geom_m = K.patch_geom(batch, 1, (n_patches,), (f,), (n_patches * f, n_patches * f, 1), token_idx=sel.drop)
    box = {}

    def run(tp: T.Tape, pv: T.Var):  # noqa: ANN202
        loss, maxes = T.op_mse(tp, pv, tgt, geom_m, norm_target, epsilon)
        box["maxes"] = maxes
        return [loss], []

    (loss,) = T.taped_call(run, [pred.float().reshape(batch * n_masked, f).contiguous()], [])
    loss = loss.reshape(())
    metrics["mse_loss"] = loss.detach()
    if norm_target and box["maxes"] is not None:
        metrics["normed_target_max"], metrics["pred_max"] = box["maxes"][0], box["maxes"][1]
    return loss, metrics


def get_decoder_patch_size(image_size: tuple, n_conv_layers: int, enc_patch_size: tuple, enc_scale_factor: tuple) -> tuple:
    """Product of the stem patch sizes (reference ``mae.py:207-228``)."""
    out = (1,) * len(image_size)
    for i in range(1 + n_conv_layers):
        p = enc_patch_size if i == 0 else enc_scale_factor
        out = tuple(a * b for a, b in zip(out, p))
    return out


class DecoderEmbedding(nn.Module):
    """Frozen sin-cos table + learnable mask token of one view (reference ``mae.py:155-204``)."""

    def __init__(self, enc_grid_size: tuple, dec_embed_dim: int, add_embed_token: bool) -> None:
        super().__init__()
        self.pos_embed = get_pos_embed(embed_dim=dec_embed_dim, grid_size=enc_grid_size)
        if add_embed_token:
            raise NotImplementedError("add_embed_token=True is never used by CineMA (mae.py:403).")
        self.embed_token = None
        self.mask_token = get_tokens(embed_dim=dec_embed_dim, n_tokens=1)


def get_model(config) -> CineMA:  # noqa: ANN001
    """Same config mapping as the reference ``get_model`` (``mae.py:231-282``); ``config`` needs attribute access."""
    views = ("sax", "lax_2c", "lax_3c", "lax_4c")
    vit = get_vit_config(config.model.size)
    model = CineMA(
        image_size_dict={v: tuple(config.data.sax.patch_size if v == "sax" else config.data.lax.patch_size) for v in views},
        in_chans_dict={v: config.data.sax.in_chans if v == "sax" else config.data.lax.in_chans for v in views},
        enc_patch_size_dict={v: tuple(config.model.patch_size if v == "sax" else config.model.patch_size[:2]) for v in views},
        enc_scale_factor_dict={v: tuple(config.model.scale_factor if v == "sax" else config.model.scale_factor[:2]) for v in views},
        enc_conv_chans=list(config.model.enc_conv_chans), enc_conv_n_blocks=config.model.enc_conv_n_blocks, **vit)
    model.set_grad_ckpt(config.grad_ckpt)
    return model


class CineMA(nn.Module):
    """Cine masked autoencoder (reference ``cinema/mae/mae.py:285-642``)."""

    def __init__(self, image_size_dict: dict, in_chans_dict: dict, enc_patch_size_dict: dict, enc_scale_factor_dict: dict, enc_conv_chans: list,
                 enc_conv_n_blocks: int, enc_embed_dim: int, enc_depth: int, enc_n_heads: int, dec_embed_dim: int, dec_depth: int, dec_n_heads: int,
                 mlp_ratio: int = 4, qkv_bias: bool = True, norm_target: bool = False, cross_attn: bool = True, norm_layer: type = nn.LayerNorm,
                 norm_eps: float = 1e-5, rotary: bool = False, act_layer: type = nn.GELU, mlp_layer: type = Mlp, drop_path: float = 0.0,
                 norm: str = "layer") -> None:
        super().__init__()
        self.grad_ckpt = False
        self.norm_target = norm_target
        self.views = list(image_size_dict.keys())
        self.enc_down_dict = nn.ModuleDict({
            v: DownsampleEncoder(image_size=tuple(image_size_dict[v]), in_chans=in_chans_dict[v], patch_size=tuple(enc_patch_size_dict[v]),
                                 scale_factor=tuple(enc_scale_factor_dict[v]), conv_chans=enc_conv_chans, conv_n_blocks=enc_conv_n_blocks,
                                 embed_dim=enc_embed_dim, norm=norm) for v in self.views})
        self.enc_fusion_dict = nn.ModuleDict({
            v: MultiScaleFusion(image_size=tuple(image_size_dict[v]), patch_size=tuple(enc_patch_size_dict[v]),
                                scale_factor=tuple(enc_scale_factor_dict[v]), conv_chans=enc_conv_chans, embed_dim=enc_embed_dim,
                                norm_layer=norm_layer, norm_eps=norm_eps) for v in self.views})
        vit_kw = dict(mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, norm_layer=norm_layer, norm_eps=norm_eps, rotary=rotary, act_layer=act_layer,
                      mlp_layer=mlp_layer, drop_path=drop_path)
        self.encoder = ViTEncoder(embed_dim=enc_embed_dim, depth=enc_depth, n_heads=enc_n_heads, **vit_kw)
        self.dec_linear = Linear(enc_embed_dim, dec_embed_dim)
        self.dec_embed_dict = nn.ModuleDict({
            v: DecoderEmbedding(enc_grid_size=self.enc_down_dict[v].patch_embed.grid_size, dec_embed_dim=dec_embed_dim, add_embed_token=False)
            for v in self.views})
        self.cross_attn = cross_attn
        self.decoder = ViTDecoder(embed_dim=dec_embed_dim, depth=dec_depth, n_heads=dec_n_heads, **vit_kw)
        self.dec_patch_size_dict = {
            v: get_decoder_patch_size(image_size=tuple(image_size_dict[v]), n_conv_layers=len(enc_conv_chans),
                                      enc_patch_size=tuple(enc_patch_size_dict[v]), enc_scale_factor=tuple(enc_scale_factor_dict[v]))
            for v in self.views}
        self.pred_head_dict = nn.ModuleDict({v: Linear(dec_embed_dim, math.prod(p) * in_chans_dict[v]) for v, p in self.dec_patch_size_dict.items()})
        self.apply(init_weights)

    def set_grad_ckpt(self, enable: bool = True) -> None:
        """Accepted for API compatibility (``mae.py:444-455``); activations are kept resident in HBM, nothing is recomputed."""
        self.grad_ckpt = enable
        for v in self.views:
            self.enc_down_dict[v].set_grad_ckpt(enable)
            self.enc_fusion_dict[v].set_grad_ckpt(enable)
            self.pred_head_dict[v].set_grad_ckpt(enable)
        self.encoder.set_grad_ckpt(enable)
        self.dec_linear.set_grad_ckpt(enable)
        self.decoder.set_grad_ckpt(enable)

    # ---------------------------------------------------------------------------------------------------------
    def _check_views(self, image_dict: dict) -> list:
        views = list(image_dict.keys())
        if any(v not in self.views for v in views):
            raise ValueError(f"views {views} must be in self.input_keys {self.views}.")
        return views

    def _encode(self, tp: T.Tape, views: list, images: dict, sels: dict, grids: dict):  # noqa: ANN202
        return encode_views(self, tp, views, images, sels, grids)

    def _forward_tape(self, tp: T.Tape, images: dict, masks: dict, n_masked: dict):  # noqa: ANN202
        views = list(images.keys())
        batch = images[views[0]].shape[0]
        dev = images[views[0]].device
        grids = {v: self.enc_down_dict[v].grid_for(tuple(images[v].shape[2:])) for v in views}
        sels = {v: TokenSelection(masks[v], batch, math.prod(grids[v]), dev, n_masked=n_masked[v]) for v in views}
        x, skips_all, cls_rows, view_rows = self._encode(tp, views, images, sels, grids)

        # fusion per view, dec_linear per segment (no concatenation needed: the GEMM is row-wise)
        parts = T.op_split_rows(tp, x, [cls_rows] + [view_rows[v] for v in views])
        d = self.dec_linear.out_features
        z_cls = T.op_linear(tp, T.op_cast_bf16(tp, parts[0]), self.dec_linear.weight, self.dec_linear.bias, out_f32=True)
        z_views, fused = {}, {}
        part_of = {v: parts[i + 1] for i, v in enumerate(views)}

        def fuse(v: str) -> None:
            fused[v] = self.enc_fusion_dict[v].tape_forward(tp, skips_all[v], part_of[v], sels[v], grids[v])

        geom = lambda v: stem_geometry(self, v, images, sels)  # noqa: E731
        # per-view weights, nothing shared: the long-axis views go out as one lane group, on the long-axis stream beside the short-axis view's fusion
        T.run_in_lanes(tp, views, geom, fuse, enabled=dev.type == "cuda", beside=True)
        for v in views:  # dec_linear is SHARED by the views (one weight-gradient buffer): not a lane group
            z_views[v] = T.op_linear(tp, fused[v], self.dec_linear.weight, self.dec_linear.bias, out_f32=True)

        # decoder sequences (mae.py:569-585)
        n_keep = [sels[v].n_keep for v in views]
        n_drop = [sels[v].n_drop for v in views]
        def rows_of(total: int, off: int, n: int) -> torch.Tensor:  # row b * total + off + i for i < n: shape-only, cached
            return T.const(("rows_of", batch, total, off, n, str(dev)), lambda: (
                torch.arange(batch, dtype=torch.int32, device=dev)[:, None] * total + off + torch.arange(n, dtype=torch.int32, device=dev)[None]).reshape(-1).contiguous())

        if self.cross_attn:
            t_q, t_k = 1 + sum(n_drop), sum(n_keep)
            q_segs, k_segs, mask_rows = [T.Segment(rows_of(t_q, 0, 1), src=z_cls, grad_bf16=True)], [], {}
            offq, offk = 1, 0
            for v, nk, nd in zip(views, n_keep, n_drop):
                emb = self.dec_embed_dict[v]
                pe = emb.pos_embed.detach().reshape(-1, d)
                k_segs.append(T.Segment(rows_of(t_k, offk, nk), src=z_views[v], add=pe, add_idx=sels[v].keep_pos, grad_bf16=True))
                mask_rows[v] = rows_of(t_q, offq, nd)
                if nd > 0:
                    q_segs.append(T.Segment(mask_rows[v], src=emb.mask_token, add=pe, add_idx=sels[v].drop_pos))
                offq, offk = offq + nd, offk + nk
            x_q = T.op_assemble(tp, batch * t_q, d, q_segs, dev)
            x_k = T.op_cast_bf16(tp, T.op_assemble(tp, batch * t_k, d, k_segs, dev))
        else:
            t_q = 1 + sum(n_keep) + sum(n_drop)
            q_segs, mask_rows = [T.Segment(rows_of(t_q, 0, 1), src=z_cls, grad_bf16=True)], {}
            off = 1
            for v, nk in zip(views, n_keep):
                pe = self.dec_embed_dict[v].pos_embed.detach().reshape(-1, d)
                q_segs.append(T.Segment(rows_of(t_q, off, nk), src=z_views[v], add=pe, add_idx=sels[v].keep_pos, grad_bf16=True))
                off += nk
            for v, nd in zip(views, n_drop):
                emb = self.dec_embed_dict[v]
                mask_rows[v] = rows_of(t_q, off, nd)
                if nd > 0:
                    q_segs.append(T.Segment(mask_rows[v], src=emb.mask_token, add=emb.pos_embed.detach().reshape(-1, d), add_idx=sels[v].drop_pos))
                off += nd
            x_q, x_k = T.op_assemble(tp, batch * t_q, d, q_segs, dev), None
        xd = self.decoder.tape_forward(tp, x_q, x_k, batch)

        # heads + loss (mae.py:589-608)
        live = [v for v in views if sels[v].n_drop > 0]
        dec_parts = dict(zip(live, T.op_split_rows(tp, xd, [mask_rows[v] for v in live]))) if live else {}
        preds, loss_of, metrics = {}, {}, {}

        def head(v: str) -> None:
            img = images[v]
            chans = img.shape[1]
            patch = self.dec_patch_size_dict[v]
            stats = T.zeros(2, torch.float32, dev)
            pgeom = K.patch_geom(batch, chans, grids[v], patch, tuple(img.stride()))
            if img.is_cuda and K.LANE is None:  # metrics only: beside the head GEMM, on the (idle) weight-gradient stream; joined below
                T._wgrad_launch(lambda: K.patch_stats(img, pgeom, stats), img, stats)  # noqa: SLF001
            else:
                K.patch_stats(img, pgeom, stats)
            metrics[f"{v}_target_mean"], metrics[f"{v}_target_std"] = stats[0], stats[1]
            if v not in dec_parts:
                preds[v] = T.Var(torch.empty((0, math.prod(patch) * chans), dtype=torch.float32, device=dev), needs_grad=False)
                nan = T.Var(K.full((1,), float("nan"), torch.float32, dev), needs_grad=False)
                loss_of[v] = nan
                metrics[f"{v}_mse_loss"] = nan.data[0]
                return
            hd = self.pred_head_dict[v]
            pred = T.op_linear(tp, dec_parts[v], hd.weight, hd.bias, out_f32=True)
            geom_m = K.patch_geom(batch, chans, grids[v], patch, tuple(img.stride()), token_idx=sels[v].drop)
            loss_v, maxes = T.op_mse(tp, pred, img, geom_m, self.norm_target)
            preds[v] = pred
            loss_of[v] = loss_v
            metrics[f"{v}_mse_loss"] = loss_v.data[0]
            if maxes is not None:
                metrics[f"{v}_normed_target_max"], metrics[f"{v}_pred_max"] = maxes[0], maxes[1]

        T.run_in_lanes(tp, views, geom, head, enabled=dev.type == "cuda", beside=True)  # prediction heads + losses: per-view weights and accumulators
        preds = {v: preds[v] for v in views}
        metrics = {k: metrics[k] for v in views for k in metrics if k.startswith(f"{v}_")}
        losses = [loss_of[v] for v in views]
        loss = T.op_mean_finite(tp, losses)
        if dev.type == "cuda":
            T.join_side_stream()  # (the target statistics above: a reader of the metrics after the forward pass sees them complete)
        return loss, preds, metrics

    def draw_masks(self, image_dict: dict, enc_mask_ratio: float) -> tuple:
        """One random mask per view, drawn exactly as ``forward`` does (same recipe, same RNG consumption, view order of ``image_dict``):
        -> ({view: bool (batch, n_patches)}, {view: masked patches per sample}).  Used by recorded steps (``cinema_amd/replay.py``)."""
        views = self._check_views(image_dict)
        batch = image_dict[views[0]].shape[0]
        dev = image_dict[views[0]].device
        masks, n_masked = {}, {}
        for v in views:
            n = math.prod(self.enc_down_dict[v].grid_for(tuple(image_dict[v].shape[2:])))
            masks[v] = get_batch_random_patch_mask(batch, n, enc_mask_ratio, dev)
            n_masked[v] = 0 if enc_mask_ratio == 0 else n - int(n * (1 - enc_mask_ratio))
        return masks, n_masked

    def forward(self, image_dict: dict, enc_mask_ratio: float, enc_mask_dict: dict | None = None, n_masked: dict | None = None):  # noqa: ANN201
        """Reference contract (``mae.py:504-520``).  ``enc_mask_dict`` (additive, optional) injects fixed masks -- used by the
        parity tests so that CPU oracle and GPU path see identical masks; ``n_masked`` (with it) gives the masked count per view
        so that it is not read back from the device."""
        views = self._check_views(image_dict)
        batch = image_dict[views[0]].shape[0]
        dev = image_dict[views[0]].device
        images = {v: image_dict[v].float().contiguous() for v in views}
        if enc_mask_dict is not None:
            masks = {v: enc_mask_dict[v].to(device=dev, dtype=torch.bool) for v in views}
            n_masked = {v: (None if n_masked is None else n_masked[v]) for v in views}  # None: count read back once
        else:
            masks, n_masked = self.draw_masks(images, enc_mask_ratio)

        out: dict = {}

        def run(tp: T.Tape):  # noqa: ANN202
            T.begin_stochastic(self, dev)  # drop_path > 0 only (the pre-training recipe uses 0)
            loss, preds, metrics = self._forward_tape(tp, images, masks, n_masked)
            out["views"], out["metric_keys"] = list(preds), list(metrics)
            return [loss], [p.data for p in preds.values()] + list(metrics.values())

        res = T.taped_call(run, [], T.trainable_params(self))
        loss = res[0].reshape(())
        n_v = len(out["views"])
        pred_dict = {}
        for v, p in zip(out["views"], res[1:1 + n_v]):
            pred_dict[v] = p.reshape(batch, -1, p.shape[-1])
        metrics = dict(zip(out["metric_keys"], [m.reshape(()) for m in res[1 + n_v:]]))
        metrics["loss"] = loss.detach()
        return loss, pred_dict, {v: masks[v] for v in views}, metrics

    def feature_forward(self, image_dict: dict) -> dict:
        """{"cls": (b, 1, E), view: (b, n_patches, E)} without masking (reference ``mae.py:457-502``)."""
        views = self._check_views(image_dict)
        batch = image_dict[views[0]].shape[0]
        dev = image_dict[views[0]].device
        images = {v: image_dict[v].float().contiguous() for v in views}

        def run(tp: T.Tape):  # noqa: ANN202
            grids = {v: self.enc_down_dict[v].grid_for(tuple(images[v].shape[2:])) for v in views}
            sels = {v: TokenSelection(None, batch, math.prod(grids[v]), dev) for v in views}
            x, skips_all, cls_rows, view_rows = self._encode(tp, views, images, sels, grids)
            parts = T.op_split_rows(tp, x, [cls_rows] + [view_rows[v] for v in views])
            outs = [parts[0]]
            for i, v in enumerate(views):
                outs.append(self.enc_fusion_dict[v].tape_forward(tp, skips_all[v], parts[i + 1], sels[v], grids[v], out_f32=True))
            return outs, []

        res = T.taped_call(run, [], T.trainable_params(self))
        return {k: r.reshape(batch, -1, r.shape[-1]) for k, r in zip(["cls", *views], res)}

    @classmethod
    def from_pretrained(cls, model_path: str | Path | None = None, config_path: str | Path | None = None, **kwargs) -> CineMA:  # noqa: ANN003
        """Load released weights (``pretrained/cinema.safetensors`` + ``config.yaml``, reference ``mae.py:614-642``).

        With no paths the files are fetched from the HF hub like the reference; on an air-gapped box pass local paths.
        """
        import yaml
        from safetensors.torch import load_file

        if model_path is None or config_path is None:
            from huggingface_hub import hf_hub_download

            model_path = model_path or hf_hub_download(repo_id="mathpluscode/CineMA", filename="pretrained/cinema.safetensors", **kwargs)
            config_path = config_path or hf_hub_download(repo_id="mathpluscode/CineMA", filename="pretrained/config.yaml", **kwargs)
        from cinema_amd.config import to_config

        with open(config_path, encoding="utf-8") as f:
            config = to_config(yaml.safe_load(f))
        model = get_model(config)
        model.load_state_dict(load_file(str(model_path)))
        return model


__all__ = ["CineMA", "DecoderEmbedding", "get_batch_random_patch_mask", "get_decoder_patch_size", "get_model", "patchify"]
