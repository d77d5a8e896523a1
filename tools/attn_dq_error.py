"""Where does the bf16 error of attn.q.weight's gradient come from at ViT-Large depth?  (encoder.blocks.22.attn.q.weight 7.9 % rel-L2.)
One forward + backward of the full-depth model of the parity test (24 + 8 blocks, 1024 / 512 channels, SAX 96 x 96 x 8 + one long-axis view, batch 2), the
operands of every encoder block's attention backward captured (tape.ATTN_CAPTURE); for the chosen blocks the kernel's dQ / dK / dV and the q-weight gradient they
imply are compared with a float64 evaluation of the same operands, and with float64 evaluations in which ONE quantity at a time is rounded the way the kernel
rounds it.  Dev tool (GPU box): python tools/attn_dq_error.py [block ...]"""
from __future__ import annotations

import sys

import torch

sys.path.insert(0, ".")
from cinema_amd import CineMA  # noqa: E402
from cinema_amd import tape as T  # noqa: E402
from cinema_amd.vit import get_vit_config  # noqa: E402

DEV = "cuda"


def rel(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


def bf(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).double()


def analyse(cap: dict, name: str) -> None:
    b, h = cap["batch"], cap["heads"]
    qkv, o16, do16, x16 = cap["qkv"], cap["o"], cap["do"], cap["x"]
    c = qkv.shape[1] // 3
    t, hd = qkv.shape[0] // b, c // h
    scale = hd ** -0.5

    def heads(m: torch.Tensor) -> torch.Tensor:  # [b*t, c] -> [b, h, t, hd]
        return m.reshape(b, t, h, hd).permute(0, 2, 1, 3).double()

    def rows(m: torch.Tensor) -> torch.Tensor:
        return m.permute(0, 2, 1, 3).reshape(b * t, c)

    q, k, v = heads(qkv[:, :c]), heads(qkv[:, c:2 * c]), heads(qkv[:, 2 * c:])
    do, o_bf = heads(do16), heads(o16)
    s = (q @ k.transpose(-1, -2)) * scale
    p = torch.softmax(s, dim=-1)
    o = p @ v                                    # exact output of these bf16 operands
    dp = do @ v.transpose(-1, -2)
    x = x16.double()
    kc = k - k.mean(dim=2, keepdim=True)         # keys without their per-head mean (dS sums to zero over the keys: the mean cannot matter in exact arithmetic)

    def dq_of(pm: torch.Tensor, delta: torch.Tensor, round_ds: bool, keys: torch.Tensor) -> torch.Tensor:
        ds = pm * (dp - delta)
        if round_ds:
            ds = bf(ds)
        return (ds @ keys) * scale

    delta_exact = (do * o).sum(-1, keepdim=True)
    delta_bf16_o = (do * o_bf).sum(-1, keepdim=True)   # what the kernel forms: rowsum(dO o O) with the stored bf16 O
    ref = dq_of(p, delta_exact, False, k)
    got = heads(cap["dqkv"][:, :c])
    gw_ref = rows(ref).t() @ x

    def line(tag: str, dq: torch.Tensor) -> None:
        print(f"  {tag:58s} dQ rel-L2 {rel(dq, ref):9.3e}   q.weight gradient rel-L2 {rel(rows(dq).t() @ x, gw_ref):9.3e}")

    print(f"{name}: t = {t}, heads = {h}, head_dim = {hd}; |mean key| / |key - mean| = {float(k.mean(2).norm() * (t ** 0.5) / kc.norm()):.2f}; "
          f"max softmax probability {float(p.max()):.3f}, mean row entropy / log t = {float(-(p * p.clamp_min(1e-300).log()).sum(-1).mean() / torch.log(torch.tensor(float(t)))):.3f}")
    line("kernel", got)
    line("float64, dS rounded to bf16", dq_of(p, delta_exact, True, k))
    line("float64, P rounded to bf16", dq_of(bf(p), delta_exact, False, k))
    line("float64, delta from the stored bf16 O", dq_of(p, delta_bf16_o, False, k))
    line("float64, all three roundings", dq_of(bf(p), delta_bf16_o, True, k))
    line("all three roundings, keys centred per head", dq_of(bf(p), delta_bf16_o, True, kc))
    line("dS rounded, keys centred", dq_of(p, delta_exact, True, kc))
    line("delta from bf16 O, keys centred", dq_of(p, delta_bf16_o, False, kc))
    line("bf16 output of dQ only (exact otherwise)", bf(ref))


def main() -> None:
    blocks = [int(a) for a in sys.argv[1:]] or [22, 12, 0]
    views = ["sax", "lax_2c"]
    kw = dict(image_size_dict={"sax": (96, 96, 8), "lax_2c": (96, 96)}, in_chans_dict=dict.fromkeys(views, 1), enc_patch_size_dict={"sax": (4, 4, 1), "lax_2c": (4, 4)},
              enc_scale_factor_dict={"sax": (2, 2, 1), "lax_2c": (2, 2)}, enc_conv_chans=[64, 128], enc_conv_n_blocks=2, **get_vit_config("large"))
    torch.manual_seed(11)
    model = CineMA(**kw).to(DEV)
    g = torch.Generator().manual_seed(13)
    images = {v: torch.rand(2, 1, *kw["image_size_dict"][v], generator=g).to(DEV) for v in views}
    T.ATTN_CAPTURE = []
    loss, _, _, _ = model(images, 0.75)
    loss.backward()
    torch.cuda.synchronize()
    caps, T.ATTN_CAPTURE = T.ATTN_CAPTURE, None
    depth = len(model.encoder.blocks)
    enc = [c for c in caps][:depth]  # backward order: block depth-1 first
    print(f"loss {float(loss):.5f}; {len(caps)} attention backward passes captured (encoder depth {depth})")
    for blk in blocks:
        analyse(enc[depth - 1 - blk], f"encoder.blocks.{blk}")


if __name__ == "__main__":
    main()
