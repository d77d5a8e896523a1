"""ConvUNetR: conv stem + shared ViT encoder + UNet-style conv decoder (interface of the reference
``cinema/segmentation/convunetr.py:25-542``).

The encoder half (stem, token assembly, ViT) is the MAE path's (``convvit.encode_views`` on all tokens, no mask).  The decoder
runs on channels-last rows: ``ConvResBlock`` = fused LayerNorm+GELU kernels + two im2col / MFMA-GEMM convs with the shortcut as the
second GEMM's fp32 residual; ``UpsampleDecoder.up`` (k == s transposed conv) = one GEMM + a patch scatter that also adds the skip.
"""

from __future__ import annotations

import math
from pathlib import Path

import torch
from torch import nn

from cinema_amd import hip as K
from cinema_amd import tape as T
from cinema_amd.conv import Conv2d, Conv3d, ConvResBlock, ConvTranspose2d, ConvTranspose3d, Volume, _CkptFlag
from cinema_amd.convvit import DownsampleEncoder, TokenSelection, encode_views, load_pretrain_weights
from cinema_amd.vit import Mlp, ViTEncoder, get_vit_config, init_weights


class UpsampleDecoder(nn.Module, _CkptFlag):
    """Up-sampling conv decoder (reference ``convunetr.py:25-106``): per level ``up`` (k == s transposed conv) + ``n_blocks`` ConvResBlocks."""

    def __init__(self, n_dims: int, chans: tuple, patch_size, scale_factor, norm: str, kernel_size: int = 3, n_blocks: int = 2,  # noqa: ANN001
                 dropout: float = 0.0) -> None:
        if n_dims not in {2, 3}:
            raise ValueError(f"Invalid n_dims, must be 2 or 3, got {n_dims}.")
        super().__init__()
        deconv_cls = ConvTranspose2d if n_dims == 2 else ConvTranspose3d
        self.blocks = nn.ModuleList()
        for i, ch in enumerate(chans[::-1]):
            block = nn.Module()
            up_kernel = patch_size if i == len(chans) - 1 else scale_factor
            out_chans = chans[-i - 2] if i < len(chans) - 1 else ch
            block.up = deconv_cls(ch, out_chans, kernel_size=up_kernel, stride=up_kernel)
            block.conv = nn.ModuleList([ConvResBlock(n_dims=n_dims, in_chans=out_chans, out_chans=out_chans, dropout=dropout,
                                                     kernel_size=kernel_size, norm=norm) for _ in range(n_blocks)])
            self.blocks.append(block)

    def set_grad_ckpt(self, enable: bool = True) -> None:
        self.grad_ckpt = enable
        for block in self.blocks:
            block.up.set_grad_ckpt(enable)
            for conv in block.conv:
                conv.set_grad_ckpt(enable)

    def tape_forward(self, tp: T.Tape, embeddings: list) -> Volume:
        """``embeddings``: Volumes (or None where a level has no skip) from fine to coarse, consumed from the end like the reference."""
        embeddings = list(embeddings)
        x = embeddings.pop()
        for block in self.blocks:
            skip = embeddings.pop()
            up = block.up
            y, sp = T.op_conv_transpose(tp, T.op_cast_bf16(tp, x.var), x.batch, x.spatial, up.weight, up.bias, skip=None if skip is None else skip.var)
            x = Volume(y, x.batch, sp, up.weight.shape[1])
            for conv in block.conv:
                x = conv.tape_forward(tp, x)
        return x


def check_conv_unetr_enc_dec_compatiblity(enc_patch_size: tuple, enc_scale_factor: tuple, enc_n_conv_layers: int, dec_depth: int,
                                          dec_patch_size: tuple, dec_scale_factor: tuple) -> tuple:
    """(n_layers_wo_skip, n_downsample_layers) or ``ValueError`` (reference ``convunetr.py:109-161``)."""
    if enc_n_conv_layers >= dec_depth:
        raise ValueError(f"enc_n_conv_layers {enc_n_conv_layers} must be less than dec_depth {dec_depth}.")
    if any(f < s for f, s in zip(enc_patch_size, dec_patch_size)):
        raise ValueError(f"enc_patch_size {enc_patch_size} must be greater than dec_patch_size {dec_patch_size}.")
    enc_patch_size, enc_scale_factor = tuple(enc_patch_size), tuple(enc_scale_factor)
    dec_patch_size, dec_scale_factor = tuple(dec_patch_size), tuple(dec_scale_factor)
    enc_factor = enc_patch_size
    for _ in range(enc_n_conv_layers):
        enc_factor = tuple(f * s for f, s in zip(enc_factor, enc_scale_factor))
    dec_factor = dec_patch_size
    n_layers_wo_skip = n_downsample_layers = None
    for i in range(dec_depth):
        if dec_factor == enc_patch_size:
            n_layers_wo_skip = i
        if dec_factor == enc_factor:
            n_downsample_layers = dec_depth - 1 - i
        dec_factor = tuple(f * s for f, s in zip(dec_factor, dec_scale_factor))
    if n_layers_wo_skip is None:
        raise ValueError(f"enc_patch_size {enc_patch_size} must be equal to dec_patch_size {dec_patch_size} times certain number of {dec_scale_factor} .")
    if n_downsample_layers is None:
        raise ValueError(f"enc_factor {enc_factor} must be equal to dec_patch_size {dec_patch_size} times certain number of {dec_scale_factor} .")
    return n_layers_wo_skip, n_downsample_layers


def get_model(config) -> "ConvUNetR":  # noqa: ANN001
    """Same config mapping as the reference ``get_model`` (``convunetr.py:164-210``)."""

    def view_cfg(v: str):  # noqa: ANN202
        if v == "sax":
            return config.data.sax
        if hasattr(config.data, "lax"):
            return config.data.lax
        return config.data[v]

    views = [config.model.views] if isinstance(config.model.views, str) else list(config.model.views)
    vit = get_vit_config(config.model.convunetr.size)
    ndim = {v: 3 if v == "sax" else 2 for v in views}
    c = config.model.convunetr
    model = ConvUNetR(image_size_dict={v: tuple(view_cfg(v).patch_size) for v in views}, in_chans_dict={v: view_cfg(v).in_chans for v in views},
                      out_chans=config.model.out_chans, enc_patch_size_dict={v: tuple(c.enc_patch_size[:n]) for v, n in ndim.items()},
                      enc_scale_factor_dict={v: tuple(c.enc_scale_factor[:n]) for v, n in ndim.items()}, enc_conv_chans=list(c.enc_conv_chans),
                      enc_conv_n_blocks=c.enc_conv_n_blocks, enc_embed_dim=vit["enc_embed_dim"], enc_depth=vit["enc_depth"],
                      enc_n_heads=vit["enc_n_heads"], dec_chans=tuple(c.dec_chans),
                      dec_patch_size_dict={v: tuple(c.dec_patch_size[:n]) for v, n in ndim.items()},
                      dec_scale_factor_dict={v: tuple(c.dec_scale_factor[:n]) for v, n in ndim.items()}, dropout=c.dropout, drop_path=c.drop_path)
    model.set_grad_ckpt(config.grad_ckpt)
    return model


class ConvUNetR(nn.Module):
    """Segmentation model (reference ``cinema/segmentation/convunetr.py:213-542``): logits (batch, out_chans, *image_size) per view."""

    def __init__(self, image_size_dict: dict, in_chans_dict: dict, out_chans: int, enc_patch_size_dict: dict, enc_scale_factor_dict: dict,
                 enc_conv_chans: list, enc_conv_n_blocks: int, enc_embed_dim: int, enc_depth: int, enc_n_heads: int, dec_chans: tuple,
                 dec_patch_size_dict: dict, dec_scale_factor_dict: dict, dec_kernel_size: int = 3, mlp_ratio: int = 4, qkv_bias: bool = True,
                 norm_layer: type = nn.LayerNorm, norm_eps: float = 1e-5, rotary: bool = False, act_layer: type = nn.GELU, mlp_layer: type = Mlp,
                 dropout: float = 0.0, drop_path: float = 0.0, norm: str = "layer") -> None:
        super().__init__()
        self.grad_ckpt = False
        self.views = list(image_size_dict.keys())
        for v in self.views:
            if len(image_size_dict[v]) not in {2, 3}:
                raise ValueError(f"Invalid image_size for {v}, must be 2D or 3D, got {image_size_dict[v]}.")
        wo_skip, n_down = [], []
        for v in self.views:
            a, b = check_conv_unetr_enc_dec_compatiblity(enc_patch_size=enc_patch_size_dict[v], enc_scale_factor=enc_scale_factor_dict[v],
                                                         enc_n_conv_layers=len(enc_conv_chans), dec_depth=len(dec_chans),
                                                         dec_patch_size=dec_patch_size_dict[v], dec_scale_factor=dec_scale_factor_dict[v])
            wo_skip.append(a)
            n_down.append(b)
        if len(set(wo_skip)) != 1:
            raise ValueError(f"n_layers_wo_skip_list {wo_skip} must be the same for all views.")
        if len(set(n_down)) != 1:
            raise ValueError(f"n_downsample_layers_list {n_down} must be the same for all views.")
        self.n_layers_wo_skip = wo_skip[0]
        n_downsample_layers = n_down[0]
        self.enc_down_dict = nn.ModuleDict({
            v: DownsampleEncoder(image_size=tuple(image_size_dict[v]), in_chans=in_chans_dict[v], patch_size=tuple(enc_patch_size_dict[v]),
                                 scale_factor=tuple(enc_scale_factor_dict[v]), conv_chans=enc_conv_chans, conv_n_blocks=enc_conv_n_blocks,
                                 embed_dim=enc_embed_dim, norm=norm) for v in self.views})
        self.encoder = ViTEncoder(embed_dim=enc_embed_dim, depth=enc_depth, n_heads=enc_n_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                  norm_layer=norm_layer, norm_eps=norm_eps, rotary=rotary, act_layer=act_layer, mlp_layer=mlp_layer, drop_path=drop_path)
        self.dec_image_conv_block_dict = nn.ModuleDict()
        self.dec_down_blocks_dict = nn.ModuleDict()
        self.dec_conv_blocks_dict = nn.ModuleDict()
        self.decoder_dict = nn.ModuleDict()
        self.pred_head_dict = nn.ModuleDict()
        for v in self.views:
            nd = len(image_size_dict[v])
            res = dict(n_dims=nd, kernel_size=dec_kernel_size, dropout=dropout, act_layer=act_layer, norm=norm)
            self.dec_image_conv_block_dict[v] = ConvResBlock(in_chans=in_chans_dict[v], out_chans=dec_chans[0], **res)
            conv_cls = Conv2d if nd == 2 else Conv3d
            self.dec_down_blocks_dict[v] = nn.ModuleList([
                conv_cls(enc_embed_dim, enc_embed_dim, kernel_size=tuple(dec_scale_factor_dict[v]), stride=tuple(dec_scale_factor_dict[v]), padding="valid")
                for _ in range(n_downsample_layers)])
            self.dec_conv_blocks_dict[v] = nn.ModuleList()
            for i, ch in enumerate(enc_conv_chans):
                self.dec_conv_blocks_dict[v].append(ConvResBlock(in_chans=ch, out_chans=dec_chans[self.n_layers_wo_skip + i], **res))
            for i in range(n_downsample_layers + 1):
                self.dec_conv_blocks_dict[v].append(
                    ConvResBlock(in_chans=enc_embed_dim, out_chans=dec_chans[self.n_layers_wo_skip + len(enc_conv_chans) + i], **res))
            self.decoder_dict[v] = UpsampleDecoder(n_dims=nd, chans=tuple(dec_chans), patch_size=tuple(dec_patch_size_dict[v]),
                                                   scale_factor=tuple(dec_scale_factor_dict[v]), norm=norm)
            self.pred_head_dict[v] = conv_cls(dec_chans[0], out_chans, kernel_size=1)
        self.apply(init_weights)

    def set_grad_ckpt(self, enable: bool = True) -> None:
        """Accepted for API compatibility (``convunetr.py:407-420``); nothing is recomputed on this path."""
        self.grad_ckpt = enable
        self.encoder.set_grad_ckpt(enable)
        for v in self.views:
            self.enc_down_dict[v].set_grad_ckpt(enable)

    def forward(self, image_dict: dict) -> dict:
        """Logits (batch, out_chans, *image_size) per view, channels first like the reference."""
        rows = self.forward_rows(image_dict)
        batch = next(iter(image_dict.values())).shape[0]
        return {v: r.reshape(batch, *image_dict[v].shape[2:], -1).movedim(-1, 1).contiguous() for v, r in rows.items()}

    def forward_rows(self, image_dict: dict) -> dict:
        """The same forward with the logits left as the kernels produce them: fp32 channels-last rows [batch * prod(image_size), out_chans] per view (what the
        loss kernels read; the recorded fine-tuning step uses this entry so that no layout copy sits between the model and the loss)."""
        views = list(image_dict.keys())
        if any(v not in self.views for v in views):
            raise ValueError(f"views {views} must be in self.input_keys {self.views}.")
        batch = image_dict[views[0]].shape[0]
        dev = image_dict[views[0]].device
        images = {v: image_dict[v].float().contiguous() for v in views}

        def run(tp: T.Tape):  # noqa: ANN202
            T.begin_stochastic(self, dev)  # dropout / drop-path of the fine-tuning recipe (acdc/config.yaml:64-65): new masks per forward
            grids = {v: self.enc_down_dict[v].grid_for(tuple(images[v].shape[2:])) for v in views}
            sels = {v: TokenSelection(None, batch, math.prod(grids[v]), dev) for v in views}
            x, skips_all, _cls_rows, view_rows = encode_views(self, tp, views, images, sels, grids)
            parts = T.op_split_rows(tp, x, [view_rows[v] for v in views])  # the cls token is not decoded (convunetr.py:455-456)
            outs = []
            e = x.data.shape[1]
            for i, v in enumerate(views):
                xv = Volume(parts[i], batch, grids[v], e)  # rows are already (batch, *grid) raster, channels last
                skips_view = list(skips_all[v]) + [xv]
                for conv in self.dec_down_blocks_dict[v]:  # k == s convs on the token grid
                    ks = tuple(conv.kernel_size)
                    grid = tuple(s // k for s, k in zip(xv.spatial, ks))
                    geom = K.patch_geom(batch, xv.chans, grid, ks, xv.strides())
                    rows = T.op_patch_gather(tp, xv.var, geom)
                    y = T.op_linear(tp, rows, conv.weight, conv.bias, out_f32=True, w16=T.w_patch(conv.weight),
                                    to_param_layout=T.patch_grad_to_param(conv.weight))
                    xv = Volume(y, batch, grid, conv.out_channels)
                    skips_view.append(xv)
                img = images[v]
                c_in = img.shape[1]
                img_rows = T.Var(img.movedim(1, -1).contiguous().reshape(-1, c_in), needs_grad=False)
                emb = [self.dec_image_conv_block_dict[v].tape_forward(tp, Volume(img_rows, batch, tuple(img.shape[2:]), c_in))]
                emb += [None] * self.n_layers_wo_skip
                for j, block in enumerate(self.dec_conv_blocks_dict[v]):
                    emb.append(block.tape_forward(tp, skips_view[j]))
                y = self.decoder_dict[v].tape_forward(tp, emb)
                head = self.pred_head_dict[v]
                outs.append(T.op_linear(tp, T.op_cast_bf16(tp, y.var), head.weight, head.bias, out_f32=True))
            return outs, []

        res = T.taped_call(run, [], T.trainable_params(self))
        return dict(zip(views, res))

    @classmethod
    def from_finetuned(cls, repo_id: str | None = None, model_filename: str | None = None, config_filename: str | None = None, *,
                       model_path: str | Path | None = None, config_path: str | Path | None = None, **kwargs) -> "ConvUNetR":  # noqa: ANN003
        """Fine-tuned weights + config (reference ``convunetr.py:487-521``); pass local paths on an air-gapped box."""
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
    def from_pretrained(cls, config, freeze: bool, model_path: str | Path | None = None, **kwargs) -> "ConvUNetR":  # noqa: ANN001, ANN003
        """MAE-pretrained stem + encoder weights into a fresh segmentation model (reference ``convunetr.py:523-542``)."""
        if model_path is None:
            from huggingface_hub import hf_hub_download

            model_path = hf_hub_download(repo_id="mathpluscode/CineMA", filename="pretrained/cinema.safetensors", **kwargs)
        return load_pretrain_weights(model=get_model(config), views=config.model.views, ckpt_path=Path(model_path), freeze=freeze)
