"""Writes tests/golden/second_opinion.safetensors: inputs and the values on which the oracle's restatement of the monai arithmetic (oracle/cinema_oracle.py) and the
independent loop-style restatement (oracle/second_opinion.py) agree.  The script REFUSES to write a case on which the two differ.  Run here (CPU):
python oracle/make_golden_second_opinion.py"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch
from safetensors.torch import save_file

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import cinema_oracle as O  # noqa: E402
import second_opinion as S  # noqa: E402


def main() -> None:
    g = torch.Generator().manual_seed(2026)
    t: dict = {}
    # ---- CE + Dice: five random cases, an absent class, ignored voxels, one class only, a 2-D case
    cases = []
    for i, (b, c, sp) in enumerate([(2, 4, (5, 6, 3)), (1, 4, (7, 5, 2)), (3, 3, (4, 4, 4)), (2, 5, (3, 3, 3)), (2, 4, (9, 2, 1))]):
        cases.append((f"rand{i}", torch.randn(b, c, *sp, generator=g) * 2, torch.randint(0, c, (b, 1, *sp), generator=g)))
    lg = torch.randn(2, 4, 4, 5, 3, generator=g)
    lab = torch.randint(0, 3, (2, 1, 4, 5, 3), generator=g)            # class 3 never occurs: its Dice term is 1 - smooth / (P + smooth)
    cases.append(("absent_class", lg, lab))
    lab2 = torch.randint(0, 4, (2, 1, 4, 5, 3), generator=g)
    lab2[0, 0, :2] = -1                                               # ignored by the cross entropy, counted as background by the Dice target
    cases.append(("ignored_voxels", torch.randn(2, 4, 4, 5, 3, generator=g), lab2))
    cases.append(("all_background", torch.randn(1, 4, 3, 3, 2, generator=g), torch.zeros(1, 1, 3, 3, 2, dtype=torch.long)))
    cases.append(("two_d", torch.randn(2, 4, 6, 7, generator=g), torch.randint(0, 4, (2, 1, 6, 7), generator=g)))
    for name, logits, labels in cases:
        _, m = O.segmentation_loss_one_view(logits.double(), labels)
        s = S.segmentation_loss(logits.numpy().astype(np.float64), labels.numpy())
        for k in ("cross_entropy", "mean_dice_loss", "loss"):
            assert abs(float(m[k]) - s[k]) <= 1e-9 * max(1.0, abs(s[k])), (name, k, float(m[k]), s[k])
        t[f"seg/{name}/logits"], t[f"seg/{name}/labels"] = logits.float().contiguous(), labels.to(torch.int32).contiguous()
        t[f"seg/{name}/values"] = torch.tensor([s["cross_entropy"], s["mean_dice_loss"], s["loss"]], dtype=torch.float64)
    # ---- Zoom(keep_size) -> ScaleIntensity -> SpatialPad(end): trilinear / bicubic, zoom in / out, odd extents, identity zoom, a constant image
    tcases = [("tri_out", (8, 9, 5), 0.87, (12, 12, 6), False), ("tri_in", (7, 6, 4), 1.13, (8, 8, 4), False), ("tri_odd", (9, 7, 3), 0.75, (9, 8, 5), False),
              ("cub_out", (10, 9), 0.9, (12, 12), True), ("cub_in", (9, 11), 1.21, (12, 12), True), ("cub_odd", (7, 5), 0.8, (8, 8), True),
              ("identity", (6, 5, 4), 1.0, (8, 8, 4), False)]
    for name, size, zoom, padded, cubic in tcases:
        x = torch.rand(*size, generator=g) * 3 - 1
        got = O.input_transform(x.double(), zoom, padded, cubic)
        ref = S.input_transform(x.numpy().astype(np.float64), zoom, padded, cubic)
        assert tuple(got.shape) == ref.shape and float(np.abs(got.numpy() - ref).max()) <= 1e-9, (name, float(np.abs(got.numpy() - ref).max()))
        t[f"tf/{name}/x"], t[f"tf/{name}/y"] = x.float().contiguous(), torch.from_numpy(ref)
        t[f"tf/{name}/args"] = torch.tensor([zoom, float(cubic), *padded], dtype=torch.float64)
    x = torch.full((4, 5, 3), 2.5)
    ref = S.input_transform(x.numpy().astype(np.float64), 0.9, (6, 6, 4), False)
    got = O.input_transform(x.double(), 0.9, (6, 6, 4), False)
    assert float(np.abs(got.numpy() - ref).max()) <= 1e-9
    t["tf/constant/x"], t["tf/constant/y"], t["tf/constant/args"] = x, torch.from_numpy(ref), torch.tensor([0.9, 0.0, 6, 6, 4], dtype=torch.float64)
    out = ROOT / "tests" / "golden" / "second_opinion.safetensors"
    save_file(t, str(out))
    print(f"wrote {out} ({out.stat().st_size} bytes, {len(t)} tensors)")


if __name__ == "__main__":
    main()
