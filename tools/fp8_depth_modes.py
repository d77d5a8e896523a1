"""Gradient error of the e4m3 modes against the fp32 CPU oracle at config 5's full depth (ViT-Large 24 + 8 blocks, small spatial size): bf16 | e4m3 forward only |
+ e4m3 data gradients with PER-ROW (per-token) scales of activations and dY, bf16 weight gradients | everything e4m3 under per-tensor delayed scales.
Dev tool (GPU box): python tools/fp8_depth_modes.py"""
import json
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "oracle")
sys.path.insert(0, "tests")
from parity import mae_fp8_grad_parity  # noqa: E402

from cinema_amd import CineMA  # noqa: E402
from cinema_amd.vit import get_vit_config  # noqa: E402

views = ["sax", "lax_2c"]
kw = dict(image_size_dict={"sax": (96, 96, 8), "lax_2c": (96, 96)}, in_chans_dict=dict.fromkeys(views, 1), enc_patch_size_dict={"sax": (4, 4, 1), "lax_2c": (4, 4)},
          enc_scale_factor_dict={"sax": (2, 2, 1), "lax_2c": (2, 2)}, enc_conv_chans=[64, 128], enc_conv_n_blocks=2, **get_vit_config("large"))
torch.manual_seed(11)
sd = {k: v.detach().clone() for k, v in CineMA(**kw).state_dict().items()}
par = mae_fp8_grad_parity(kw, sd, batch=2, seed=13, device="cuda", threads=16)
for mode in ("bf16", "fp8_forward", "fp8", "fp8_wgrad"):
    m = par[mode]
    blocks = m["block_matrix_rel_l2"]
    print(f"{mode:12s} loss rel {m['loss_rel']:.2e}  grad-norm rel {m['grad_norm_rel']:.2e}  whole gradient {100 * m['whole_grad_rel_l2']:.2f} %  worst matrix "
          f"{100 * m['worst_matrix_rel_l2']['value']:.1f} % ({m['worst_matrix_rel_l2']['name']})  blocks enc0 {100 * blocks['encoder.00']:.1f} % enc23 {100 * blocks['encoder.23']:.1f} % "
          f"dec0 {100 * blocks['decoder.00']:.1f} % dec7 {100 * blocks['decoder.07']:.1f} %  e4m3 dgrad GEMMs {m['fp8_dgrad_gemms']} wgrad problems {m['fp8_wgrad_problems']}")
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "block_matrix_rel_l2"} for k, v in par.items() if isinstance(v, dict)})[:2000])
