"""First-step parity of the HIP path against the CPU oracle on identical weights, inputs and masks -- TEST INFRASTRUCTURE ONLY.

Used by ``tests/test_model_gpu.py`` / ``tests/test_seg_gpu.py`` (the real BASELINE config-2, config-4 and config-5 shapes) and by ``bench.py``'s
``cpu_baseline`` leg, which prints the results as the ``parity`` objects of its JSON line.  The product package never imports this file.
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


def seg_step_parity(kw: dict, state_dict: dict, device: str = "cuda", seed: int = 99, threads: int | None = None) -> dict:
    """BASELINE config 4 acceptance on ONE sample: ``cinema_amd`` ConvUNetR(**kw) with dropout / drop_path switched off against the oracle on the
    same ``state_dict`` and the same U[0,1) volume (labels = the intensity quantised into the classes): eval-mode logits -> argmax agreement and the
    foreground Dice between the two argmax segmentations (>= 0.995, |1 - Dice| <= 0.01), and train-mode CE + Dice loss / global gradient
    norm / worst per-tensor gradient against ``O.segmentation_loss_one_view`` + autograd."""
    from cinema_amd.segmentation.convunetr import ConvUNetR
    from cinema_amd.segmentation.train import segmentation_loss_tensors, segmentation_metrics

    if threads:
        torch.set_num_threads(threads)
    kw = dict(kw, dropout=0.0, drop_path=0.0)
    n_cls = kw["out_chans"]
    cfg = O.MAEConfig(image_size_dict=kw["image_size_dict"], in_chans_dict=kw["in_chans_dict"], enc_patch_size_dict=kw["enc_patch_size_dict"],
                      enc_scale_factor_dict=kw["enc_scale_factor_dict"], enc_conv_chans=kw["enc_conv_chans"], enc_conv_n_blocks=kw["enc_conv_n_blocks"],
                      enc_embed_dim=kw["enc_embed_dim"], enc_depth=kw["enc_depth"], enc_n_heads=kw["enc_n_heads"], dec_embed_dim=16, dec_depth=1, dec_n_heads=2)
    gen = torch.Generator().manual_seed(seed)
    image = torch.rand(1, 1, *kw["image_size_dict"]["sax"], generator=gen)
    labels = torch.clamp((image * n_cls).long(), 0, n_cls - 1)
    sd = {k: v.detach().float().cpu() for k, v in state_dict.items()}
    p = {k: v.clone().requires_grad_(not k.endswith("pos_embed")) for k, v in sd.items()}
    t0 = time.perf_counter()
    ref = O.convunetr_forward(p, cfg, tuple(kw["dec_chans"]), 1, 1, {"sax": image})["sax"]
    ref_loss, _ = O.segmentation_loss_one_view(ref, labels)
    ref_loss.backward()
    cpu_s = time.perf_counter() - t0

    model = ConvUNetR(**kw)
    model.load_state_dict(sd)
    model.to(device).eval()
    with torch.no_grad():
        got = model({"sax": image.to(device)})["sax"].float()
    ref_d = ref.detach()
    agree = float((got.argmax(1).cpu() == ref_d.argmax(1)).float().mean())
    m = segmentation_metrics(got, ref_d.argmax(1, keepdim=True).to(device), (1.0, 1.0, 10.0))
    dice = float(torch.nanmean(torch.stack([m[f"class_{k}_dice_score"] for k in range(1, n_cls)])))
    model.train()
    loss, _ = segmentation_loss_tensors(model, {"sax_image": image.to(device), "sax_label": labels.to(torch.int8).to(device)}, ["sax"], torch.device(device))
    loss.backward()
    sq_g = sq_r = 0.0
    worst, worst_small, worst_global = ("", 0.0), ("", 0.0, 0.0), ("", 0.0)
    pairs = [(k, q.grad.float().cpu(), p[k].grad) for k, q in model.named_parameters() if p[k].grad is not None and q.grad is not None]
    ref_norm = math.sqrt(sum(float(r.double().pow(2).sum()) for _, _, r in pairs))
    for k, g, r in pairs:
        sq_g += float(g.double().pow(2).sum())
        sq_r += float(r.double().pow(2).sum())
        share = float(r.norm()) / ref_norm
        err = float((g - r).norm())
        if err / ref_norm > worst_global[1]:
            worst_global = (k, err / ref_norm)
        if share < 1e-6:  # e.g. the weight of a LayerNorm over ONE channel (the raw-image block): its input is identically zero after the mean is
            continue      # removed, the gradient is rounding noise on both sides
        l2 = err / float(r.norm())
        # tensors that carry less than 1e-3 of the gradient norm (at random init: the q / k projections of the T = 3073 attention, whose softmax is
        # nearly uniform) are dominated by bf16 rounding of much larger intermediate terms: reported, bounded relative to the GLOBAL norm
        if share >= 1e-3:
            if l2 > worst[1]:
                worst = (k, l2)
        elif l2 > worst_small[1]:
            worst_small = (k, l2, share)
    return {"argmax_agreement": agree, "dice_gpu_vs_cpu_segmentation": dice, "logits_max_abs": float((got.cpu() - ref_d).abs().max()),
            "logits_abs_max_ref": float(ref_d.abs().max()), "loss": float(loss), "oracle_loss": float(ref_loss),
            "loss_rel": abs(float(loss) - float(ref_loss)) / abs(float(ref_loss)), "grad_norm_rel": abs(math.sqrt(sq_g) - math.sqrt(sq_r)) / math.sqrt(sq_r),
            "worst_grad_rel_l2": {"name": worst[0], "value": worst[1]},
            "worst_small_grad_rel_l2": {"name": worst_small[0], "value": worst_small[1], "share_of_grad_norm": worst_small[2]},
            "worst_grad_err_over_global_norm": {"name": worst_global[0], "value": worst_global[1]}, "oracle_seconds": round(cpu_s, 2)}


def mae_loss_parity(kw: dict, state_dict: dict, batch: int = 1, seed: int = 7, device: str = "cuda", fp8: bool = False, threads: int | None = None) -> dict:
    """Forward-only first-step loss of the HIP path (optionally with the e4m3 forward projections) against the oracle: the check that fits the Large
    256 x 256 x 24 shape of BASELINE config 5 into a test (the oracle's backward at that shape takes minutes, its forward seconds)."""
    from cinema_amd import CineMA
    from cinema_amd import tape as T

    if threads:
        torch.set_num_threads(threads)
    cfg = O.MAEConfig(**{k: (dict(v) if isinstance(v, dict) else v) for k, v in kw.items()})
    gen = torch.Generator().manual_seed(seed)
    images = {v: torch.rand(batch, 1, *s, generator=gen) for v, s in kw["image_size_dict"].items()}
    masks = {v: O.random_patch_mask(batch, math.prod(cfg.grid_size(v)), 0.75, gen) for v in images}
    sd = {k: v.detach().float().cpu() for k, v in state_dict.items()}
    t0 = time.perf_counter()
    with torch.no_grad():
        ref_loss, _, ref_metrics = O.mae_forward(sd, cfg, images, masks)
    cpu_s = time.perf_counter() - t0
    model = CineMA(**kw)
    model.load_state_dict(sd)
    model.to(device)
    prev, T.FP8_FORWARD = T.FP8_FORWARD, fp8
    try:
        with torch.no_grad():
            loss, _, _, metrics = model({k: v.to(device) for k, v in images.items()}, 0.75, enc_mask_dict={k: v.to(device) for k, v in masks.items()})
    finally:
        T.FP8_FORWARD = prev
    return {"loss": float(loss), "oracle_loss": float(ref_loss), "loss_rel": abs(float(loss) - float(ref_loss)) / abs(float(ref_loss)),
            "view_loss_rel": {v: abs(float(metrics[f"{v}_mse_loss"]) - float(ref_metrics[f"{v}_mse_loss"])) / abs(float(ref_metrics[f"{v}_mse_loss"])) for v in images},
            "fp8": fp8, "oracle_seconds": round(cpu_s, 2)}


def mae_fp8_grad_parity(kw: dict, state_dict: dict, batch: int = 3, seed: int = 5, device: str = "cuda", threads: int | None = None,
                        modes: tuple = ("bf16", "fp8_forward", "fp8", "fp8_wgrad"), report: tuple = ()) -> dict:
    """Gradients of the fp8 path against the ORACLE (not against this repository's bf16 path): one forward + backward of the oracle on the CPU and three
    of the HIP path on identical weights / inputs / masks - bf16, e4m3 forward only, e4m3 forward + e4m3 data gradients, and the same + e4m3 WEIGHT gradients
    (per-tensor delayed scaling: a first pass records the maxima).  The flat parameter buffers
    (``cinema_amd.optim.FlatModel``) are built BEFORE the comparison: the transposed e4m3 weight shadows the data-gradient GEMM reads exist only there, and
    the number of GEMMs that really took them is returned (``fp8_dgrad_gemms``; 0 would mean the bf16 fallback was measured).  Per mode: loss rel, gradient-norm
    rel, worst relative L2 over matrices / conv filters (dim >= 2) and over vectors, and the error of the WHOLE gradient (all tensors concatenated) over
    the oracle's gradient norm."""
    from cinema_amd import CineMA
    from cinema_amd import tape as T
    from cinema_amd.optim import FlatModel

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
    ref_loss, _, _ = O.mae_forward(p, cfg, images, masks)
    ref_loss.backward()
    cpu_s = time.perf_counter() - t0
    ref = {k: p[k].grad for k in train if p[k].grad is not None}
    ref_norm = math.sqrt(sum(float(r.double().pow(2).sum()) for r in ref.values()))

    model = CineMA(**kw)
    model.load_state_dict(sd)
    model.to(device)
    flat = FlatModel(model, 0.05)
    named = dict(model.named_parameters())
    dimg, dmask = {k: v.to(device) for k, v in images.items()}, {k: v.to(device) for k, v in masks.items()}
    calls = {"n": 0}
    orig_wt = T.w_fp8_t

    def counting_wt(weight):  # noqa: ANN001, ANN202
        hit = orig_wt(weight)
        calls["n"] += hit is not None
        return hit

    from cinema_amd import hip as K

    wg8 = {"n": 0}
    orig_wg8 = K.gemm_fp8_wgrad_grouped

    def counting_wg8(problems):  # noqa: ANN001, ANN202
        wg8["n"] += len(problems)
        return orig_wg8(problems)

    prev = (T.FP8_FORWARD, T.FP8_DGRAD, T.FP8_WGRAD)
    out = {"oracle_loss": float(ref_loss), "oracle_grad_norm": ref_norm, "oracle_seconds": round(cpu_s, 2), "batch": batch}
    T.w_fp8_t = counting_wt
    K.gemm_fp8_wgrad_grouped = counting_wg8
    try:
        for mode, (fwd8, dg8, wg) in {"bf16": (False, False, False), "fp8_forward": (True, False, False), "fp8": (True, True, False),
                                      "fp8_wgrad": (True, True, True)}.items():
            if mode not in modes:
                continue
            T.FP8_FORWARD, T.FP8_DGRAD, T.FP8_WGRAD = fwd8, dg8, wg
            if wg:  # delayed per-tensor scaling: one pass records the maxima of every site, the measured pass quantises with them (same weights and inputs)
                flat.zero_grad()
                loss, _, _, _ = model(dimg, 0.75, enc_mask_dict=dmask)
                loss.backward()
                T.fp8_step_end()
            calls["n"] = wg8["n"] = 0
            flat.zero_grad()
            loss, _, _, _ = model(dimg, 0.75, enc_mask_dict=dmask)
            loss.backward()
            sq_g = sq_e = 0.0
            named_l2 = {}
            worst_m, worst_v = ("", 0.0), ("", 0.0)
            by_depth: dict = {}  # encoder / decoder block index -> (squared error, squared reference norm) over the block's matrices
            for k, r in ref.items():
                g = named[k].grad.float().cpu()
                e = float((g - r).double().pow(2).sum())
                parts = k.split(".")
                if r.dim() >= 2 and len(parts) > 3 and parts[1] == "blocks":
                    acc = by_depth.setdefault(f"{parts[0]}.{int(parts[2]):02d}", [0.0, 0.0])
                    acc[0] += e
                    acc[1] += float(r.double().pow(2).sum())
                sq_g += float(g.double().pow(2).sum())
                sq_e += e
                rn = float(r.norm())
                if rn < 1e-6 * ref_norm:  # numerically nothing on both sides (LayerNorm over one channel)
                    continue
                l2 = math.sqrt(e) / rn
                if k in report:
                    named_l2[k] = round(l2, 5)
                if r.dim() >= 2 and l2 > worst_m[1]:
                    worst_m = (k, l2)
                if r.dim() < 2 and l2 > worst_v[1]:
                    worst_v = (k, l2)
            out[mode] = {"loss": float(loss), "loss_rel": abs(float(loss) - float(ref_loss)) / abs(float(ref_loss)),
                         "grad_norm_rel": abs(math.sqrt(sq_g) - ref_norm) / ref_norm, "whole_grad_rel_l2": math.sqrt(sq_e) / ref_norm,
                         "worst_matrix_rel_l2": {"name": worst_m[0], "value": worst_m[1]}, "worst_vector_rel_l2": {"name": worst_v[0], "value": worst_v[1]},
                         "fp8_dgrad_gemms": calls["n"], "fp8_wgrad_problems": wg8["n"], "named_rel_l2": named_l2,
                         # relative L2 error of each transformer block's matrices taken together: how the error grows from the last block (where the
                         # backward pass starts) to the first
                         "block_matrix_rel_l2": {b: round(math.sqrt(v[0] / v[1]), 5) for b, v in sorted(by_depth.items()) if v[1] > 0}}
    finally:
        T.w_fp8_t = orig_wt
        K.gemm_fp8_wgrad_grouped = orig_wg8
        T.FP8_FORWARD, T.FP8_DGRAD, T.FP8_WGRAD = prev
    return out
