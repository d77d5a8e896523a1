"""First-step parity of the HIP path against the CPU oracle on identical weights, inputs and masks -- TEST INFRASTRUCTURE ONLY.

Used by ``tests/test_model_gpu.py`` (the real BASELINE config-2 shape) and by ``bench.py``'s ``cpu_baseline`` leg, which prints the
result as the ``parity`` object of its JSON line.  The product package never imports this file.
"""

from __future__ import annotations

import math
import time

import torch

import cinema_oracle as O  # noqa: N812

# gradients compared by name (stem, encoder, fusion, decoder, tokens, head): reference parameter names (cinema/mae/mae.py:285-442)
NAMED_GRADS = (
    "enc_down_dict.sax.conv_blocks.0.conv.0.dw_conv.weight",
    "enc_down_dict.sax.patch_embed.proj.weight",
    "encoder.blocks.0.attn.kv.weight",
    "encoder.blocks.11.mlp.fc1.weight",
    "enc_fusion_dict.sax.down_convs.0.weight",
    "dec_linear.weight",
    "decoder.blocks.0.attn.q.weight",
    "decoder.blocks.7.mlp.fc2.weight",
    "dec_embed_dict.sax.mask_token",
    "pred_head_dict.lax_2c.weight",
)


def mae_step_parity(kw: dict, state_dict: dict, batch: int = 2, seed: int = 7, device: str = "cuda", threads: int | None = None) -> dict:
    """Forward + backward of ``cinema_amd.CineMA(**kw)`` on ``device`` and of the oracle on the CPU: same ``state_dict``, same U[0,1)
    images, same injected 75 % masks.  Returns relative errors (loss, per-view losses, global gradient norm, named and worst gradients)."""
    from cinema_amd import CineMA

    if threads:
        torch.set_num_threads(threads)
    cfg = O.MAEConfig(**{k: (dict(v) if isinstance(v, dict) else v) for k, v in kw.items()})
    gen = torch.Generator().manual_seed(seed)
    images = {v: torch.rand(batch, 1, *s, generator=gen) for v, s in kw["image_size_dict"].items()}
    masks = {v: O.random_patch_mask(batch, math.prod(cfg.grid_size(v)), 0.75, gen) for v in images}
    sd = {k: v.detach().float().cpu() for k, v in state_dict.items()}
    train = set(O.trainable_keys(sd))
    p = {k: v.clone().requires_grad_(k in train) for k, v in sd.items()}
    t0 = time.perf_counter()
    ref_loss, ref_pred, ref_metrics = O.mae_forward(p, cfg, images, masks)
    ref_loss.backward()
    cpu_s = time.perf_counter() - t0

    model = CineMA(**kw)
    model.load_state_dict(sd)
    model.to(device)
    loss, pred, _, metrics = model({k: v.to(device) for k, v in images.items()}, 0.75, enc_mask_dict={k: v.to(device) for k, v in masks.items()})
    loss.backward()
    named = dict(model.named_parameters())

    def rel(a: float, b: float) -> float:
        return abs(a - b) / max(abs(b), 1e-30)

    out = {"batch": batch, "loss": float(loss), "oracle_loss": float(ref_loss), "loss_rel": rel(float(loss), float(ref_loss)),
           "oracle_seconds": round(cpu_s, 2)}
    out["view_loss_rel"] = {v: rel(float(metrics[f"{v}_mse_loss"]), float(ref_metrics[f"{v}_mse_loss"])) for v in images}
    out["pred_max_abs"] = max(float((pred[v].float().cpu() - ref_pred[v]).abs().max()) for v in images)
    sq_g = sq_r = 0.0
    worst = ("", 0.0)
    worst_max = ("", 0.0)
    per_name = {}
    for k in train:
        r = p[k].grad
        if r is None:
            continue
        g = named[k].grad.float().cpu()
        sq_g += float(g.double().pow(2).sum())
        sq_r += float(r.double().pow(2).sum())
        l2 = float((g - r).norm() / r.norm().clamp_min(1e-30))
        mx = float((g - r).abs().max() / r.abs().max().clamp_min(1e-30))
        if l2 > worst[1]:
            worst = (k, l2)
        if mx > worst_max[1]:
            worst_max = (k, mx)
        if k in NAMED_GRADS:
            per_name[k] = {"rel_l2": l2, "max_abs_over_max": mx}
    out["grad_norm"], out["oracle_grad_norm"] = math.sqrt(sq_g), math.sqrt(sq_r)
    out["grad_norm_rel"] = rel(out["grad_norm"], out["oracle_grad_norm"])
    out["named_grads"] = per_name
    out["worst_grad_rel_l2"] = {"name": worst[0], "value": worst[1]}
    out["worst_grad_max_abs_over_max"] = {"name": worst_max[0], "value": worst_max[1]}
    out["grad_rel"] = max(v["rel_l2"] for v in per_name.values()) if per_name else worst[1]
    return out
