"""Race hunt, two ranks on one GPU over gloo (the set-up of tests/test_ddp_gpu.py part a): the same eager forward + backward + overlapped gradient exchange N times;
tensors whose reduced gradient differs from the first iteration's by more than rounding noise are reported per rank.
   CINEMA_SIDE_STREAMS=2 python tools/side_stream_race_ddp.py [iterations]"""
import math
import os
import sys
from pathlib import Path

import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def worker(rank: int, world: int, port: int, n: int) -> None:
    for p in (str(ROOT), str(ROOT / "oracle")):
        sys.path.insert(0, p)
    import cinema_oracle as O  # noqa: N812
    from cinema_amd import CineMA
    from cinema_amd.ddp import GradientSynchronizer, ddp_setup
    from cinema_amd.optim import FlatModel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    torch.cuda.set_device(0)
    ddp_setup(rank, world, port=port, backend="gloo")
    views = ["sax", "lax_2c"]
    kw = dict(image_size_dict={"sax": (64, 64, 8), "lax_2c": (64, 64)}, in_chans_dict=dict.fromkeys(views, 1), enc_patch_size_dict={"sax": (4, 4, 1), "lax_2c": (4, 4)},
              enc_scale_factor_dict={"sax": (2, 2, 1), "lax_2c": (2, 2)}, enc_conv_chans=[64, 128], enc_conv_n_blocks=1, enc_embed_dim=256, enc_depth=2, enc_n_heads=4,
              dec_embed_dim=128, dec_depth=2, dec_n_heads=4)
    cfg = O.MAEConfig(**kw)
    gen = torch.Generator().manual_seed(3)
    images = {v: torch.rand(4, 1, *s, generator=gen) for v, s in kw["image_size_dict"].items()}
    masks = {v: O.random_patch_mask(4, math.prod(cfg.grid_size(v)), 0.75, gen) for v in images}
    half = slice(2 * rank, 2 * rank + 2)
    torch.manual_seed(100)
    model = CineMA(**kw).to("cuda")
    flat = FlatModel(model, 0.05)
    sync = GradientSynchronizer(world)
    sync.min_early = 1 << 12
    sync.attach(flat)
    names = {flat.offsets[id(p)][0]: k for k, p in model.named_parameters() if id(p) in flat.offsets}
    img = {v: images[v][half].cuda() for v in images}
    msk = {v: masks[v][half].cuda() for v in images}
    ref, bad = None, 0
    for it in range(n):
        flat.zero_grad()
        loss, _, _, _ = model(img, 0.75, enc_mask_dict=msk)
        sync.arm(True)
        loss.backward()
        sync.all_reduce()
        torch.cuda.synchronize()
        g = flat.flat_grad.clone()
        if ref is None:
            ref = g
            continue
        diff = (g - ref).abs()
        if float(diff.max()) > 1e-3 * float(ref.abs().max()):
            bad += 1
            worst = []
            for p in model.parameters():
                a, b = flat.offsets[id(p)]
                d, r = float(diff[a:b].max()), float(ref[a:b].abs().max())
                if d > 1e-3 * max(r, 1e-12):
                    worst.append((names[a], round(d / max(r, 1e-12), 4)))
            print(f"rank {rank} iteration {it}: {len(worst)} tensors differ: {worst[:10]}", flush=True)
    print(f"rank {rank}: RACE HUNT", "clean" if bad == 0 else f"{bad} of {n - 1} iterations differ", "early collectives", sync.n_early_last, flush=True)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    sys.path.insert(0, str(ROOT))
    from cinema_amd.ddp import get_free_port

    mp.spawn(worker, args=(2, get_free_port(), int(sys.argv[1]) if len(sys.argv) > 1 else 40), nprocs=2, join=True)
