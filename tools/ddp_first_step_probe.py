import math, os, sys, tempfile
from pathlib import Path
import torch, torch.multiprocessing as mp
ROOT = Path("/root/repo")
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
def main():
    from cinema_amd.ddp import get_free_port
    import test_ddp_gpu as TD
    tmp = tempfile.mkdtemp()
    mp.spawn(TD._worker, args=(2, get_free_port(), tmp), nprocs=2, join=True)
    a0, a1, full = (torch.load(f"{tmp}/{n}.pt") for n in ("a0", "a1", "full"))
    rel = float((a0["grad"] - full["grad"]).norm() / full["grad"].norm())
    print("rel", rel, "ranks equal", torch.equal(a0["grad"], a1["grad"]))
    if rel > 2e-2:
        from cinema_amd import CineMA
        from cinema_amd.optim import FlatModel
        model = CineMA(**TD._kwargs())
        flat = FlatModel(model, 0.05)
        d = (a0["grad"] - full["grad"])
        out = []
        for k, p in model.named_parameters():
            if id(p) in flat.offsets:
                a, b = flat.offsets[id(p)]
                fn = float(full["grad"][a:b].norm())
                e = float(d[a:b].norm())
                if e > 0.05 * max(fn, 1e-12) and e > 1e-3 * float(full["grad"].norm()):
                    out.append((k, round(e / max(fn, 1e-12), 3), round(float(a0["grad"][a:b].norm()) / max(fn, 1e-12), 3)))
        print("BAD tensors (name, err/ref, got/ref):", out[:20])
if __name__ == "__main__":
    main()
