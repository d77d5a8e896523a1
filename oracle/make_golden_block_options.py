"""Golden vectors for the transformer-block OPTIONS that no shipped CineMA config sets but the reference constructors accept (``cinema/vit.py:446-609``: ``init_values`` =
timm LayerScale, ``qk_norm``) and for ``PatchEmbed(dynamic_img_pad=True)`` (``vit.py:332-337``), generated from the upstream reference (runs ONLY where /root/reference
exists):  python oracle/make_golden_block_options.py  ->  tests/golden/block_options.safetensors

Per case: the block's parameters (randomised, LayerScale gammas and the q / k norms included), an input (and keys for the cross-attention case), the output, and the
gradients of every parameter and of the input under loss = sum(out * w) with a fixed random w.  timm is not installed: ``LayerScale`` comes from the stand-in of
``oracle/ref_shim.py`` (x * gamma, gamma = init_values * ones - timm 1.0.15).  Data only."""
from __future__ import annotations

import sys
from pathlib import Path

import torch
from safetensors.torch import save_file
from torch import nn

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import ref_shim  # noqa: E402

ref_shim.install()
from cinema.vit import Block, PatchEmbed  # noqa: E402
from timm.layers import Mlp  # noqa: E402  (the stand-in)

OUT = HERE.parent / "tests" / "golden"
CASES = {  # name -> (dim, heads, tokens q, tokens k or 0, block kwargs)
    "layerscale_hd16": (64, 4, 24, 0, dict(init_values=0.1)),
    "layerscale_hd64_cross": (128, 2, 40, 56, dict(init_values=0.5)),
    "qknorm_hd16": (64, 4, 24, 0, dict(qk_norm=True)),
    "qknorm_hd64_cross": (128, 2, 40, 56, dict(qk_norm=True)),
    "qknorm_layerscale_hd32": (128, 4, 33, 0, dict(qk_norm=True, init_values=0.2)),
}


def main() -> None:
    torch.set_num_threads(8)
    t: dict = {}
    for name, (dim, heads, tq, tk, kw) in CASES.items():
        torch.manual_seed(sum(map(ord, name)))
        blk = Block(dim=dim, n_heads=heads, mlp_ratio=4, norm_layer=nn.LayerNorm, norm_eps=1e-6, drop_path=0.0, qkv_bias=True, rotary=False, act_layer=nn.GELU,
                    mlp_layer=Mlp, **kw)
        with torch.no_grad():
            for n, p in blk.named_parameters():  # every parameter random, so that no gradient is trivially zero (LayerNorm weights ~ 1, LayerScale gammas spread)
                if n.endswith("norm.weight") or n.endswith("norm1.weight") or n.endswith("norm2.weight"):
                    p.copy_(1.0 + 0.2 * torch.randn_like(p))
                elif "gamma" in n:
                    p.copy_(p * (1.0 + 0.5 * torch.rand_like(p)))
                elif p.dim() == 1:
                    p.copy_(0.1 * torch.randn_like(p))
                else:
                    p.copy_(torch.randn_like(p) * p.shape[1] ** -0.5)
        blk.eval()
        q = torch.randn(2, tq, dim, requires_grad=True)
        k = torch.randn(2, tk, dim, requires_grad=True) if tk else None
        out = blk(q, k)
        w = torch.randn_like(out)
        (out * w).sum().backward()
        for n, p in blk.named_parameters():
            t[f"{name}/param/{n}"] = p.detach().clone()
            t[f"{name}/grad/{n}"] = p.grad.detach().clone()
        t[f"{name}/q"], t[f"{name}/dq"], t[f"{name}/out"], t[f"{name}/w"] = q.detach().clone(), q.grad.clone(), out.detach().clone(), w
        if k is not None:
            t[f"{name}/k"], t[f"{name}/dk"] = k.detach().clone(), k.grad.clone()
    # PatchEmbed with dynamic padding - the reference hands its per-AXIS pad pairs to
    # F.pad, whose first pair belongs to the LAST axis (the fixture pins that behaviour); embed_dim 32
    # (cases whose pad amounts agree under that reversal run; "pad3d_mixed" does not and the reference's own patchify raises - pinned as such)
    for name, size, patch in (("pad2d", (30, 30), (4, 4)), ("pad2d_aniso", (10, 14), (4, 8)), ("pad3d", (9, 9, 9), (4, 4, 4)), ("pad3d_mixed", (10, 13, 7), (4, 4, 2))):
        torch.manual_seed(len(name) + size[0])
        try:
            pe = PatchEmbed(image_size=size, patch_size=patch, in_chans=2, embed_dim=32, dynamic_img_pad=True)
            x = torch.randn(2, 2, *size)
            y = pe(x)
        except Exception as e:  # noqa: BLE001  (a pad that leaves an axis indivisible makes the reference's own patchify fail: recorded as such)
            t[f"{name}/raises"] = torch.tensor([1.0])
            print(name, "reference raises:", type(e).__name__, str(e)[:120])
            continue
        t[f"{name}/x"], t[f"{name}/y"] = x, y.detach().clone()
        t[f"{name}/weight"], t[f"{name}/bias"] = pe.proj.weight.detach().clone(), pe.proj.bias.detach().clone()
    save_file({k: v.contiguous() for k, v in t.items()}, str(OUT / "block_options.safetensors"))
    print(len(t), "tensors ->", OUT / "block_options.safetensors")


if __name__ == "__main__":
    main()
