"""GPU input pipeline against the oracle restatement of the loader transforms (cinema/mae/pretrain.py:157-200)."""

from __future__ import annotations

import pytest
import torch

pytestmark = pytest.mark.gpu

import cinema_oracle as O  # noqa: E402
from cinema_amd import hip as K  # noqa: E402

DEV = "cuda"


@pytest.mark.parametrize(("size", "padded", "zoom", "cubic"), [
    ((40, 36, 7), (48, 48, 8), 1.0, False), ((40, 36, 7), (48, 48, 8), 0.9, False), ((40, 36, 7), (40, 36, 7), 1.1, False),
    ((50, 44), (64, 64), 1.0, True), ((50, 44), (64, 64), 0.93, True), ((50, 44), (50, 44), 1.07, True), ((33, 20, 5), (33, 20, 6), 1.05, False)])
def test_zoom_scale_pad_vs_oracle(size: tuple, padded: tuple, zoom: float, cubic: bool) -> None:
    g = torch.Generator().manual_seed(3)
    x = torch.rand(size, generator=g) * 300.0 - 20.0  # raw intensities, not yet in [0, 1]
    want = O.input_transform(x, zoom, padded, cubic)
    dst = torch.full(padded, 7.0, device=DEV)
    K.zoom_scale_pad(x.to(DEV), (zoom,) * len(size), dst, cubic=cubic)
    got = dst.cpu()
    assert got.shape == want.shape and float(got.min()) == 0.0 and float(got.max()) == pytest.approx(1.0, abs=1e-6)
    assert float((got - want).abs().max()) <= (1e-6 if zoom == 1.0 else 2e-5), float((got - want).abs().max())
    const = torch.full(size, 3.0)
    K.zoom_scale_pad(const.to(DEV), (1.0,) * len(size), dst, cubic=cubic)
    assert float(dst.abs().max()) == 0.0  # a constant image scales to zeros (monai rescale_array with minv = 0)


def test_pipeline_double_buffering_and_model_ready_batches() -> None:
    from cinema_amd.data import GpuInputPipeline

    sizes = {"sax": (32, 32, 4), "lax_2c": (32, 32)}
    pipe = GpuInputPipeline(sizes, device=DEV, prob=0.5, seed=1)
    g = torch.Generator().manual_seed(0)

    def subjects(n: int) -> list:
        return [{"sax": torch.rand(30, 28, 3, generator=g) * 100, "lax_2c": torch.rand(32, 30, generator=g) * 50} for _ in range(n)]

    pipe.submit(subjects(3))
    pipe.submit(subjects(3))  # the second upload is in flight while the first batch is transformed
    a, b = pipe.get(), pipe.get()
    for batch in (a, b):
        assert batch["sax"].shape == (3, 1, 32, 32, 4) and batch["lax_2c"].shape == (3, 1, 32, 32)
        for v in batch:
            assert float(batch[v].min()) == 0.0 and float(batch[v].amax()) == pytest.approx(1.0, abs=1e-6)
        assert float(batch["sax"][:, :, 30:].abs().max()) == 0.0 and float(batch["lax_2c"][:, :, :, 30:].abs().max()) == 0.0  # end padding
    assert not torch.equal(a["sax"], b["sax"])
    with pytest.raises(RuntimeError):
        pipe.get()
    with pytest.raises(ValueError, match="exceeds the padded size"):
        pipe.submit([{"sax": torch.rand(40, 28, 3)}])


def test_zoom_scale_pad_vs_the_pinned_second_opinion_vectors() -> None:
    """The HIP input transform against ``tests/golden/second_opinion.safetensors`` (values on which the oracle and the independent loop-style statement of monai's
    Zoom / ScaleIntensity / SpatialPad agree; oracle/make_golden_second_opinion.py): trilinear / bicubic, zoom in / out, odd extents, identity, constant image."""
    from conftest import load_golden

    g = load_golden("second_opinion.safetensors")
    for name in sorted({k.split("/")[1] for k in g if k.startswith("tf/")}):
        x, want, args = g[f"tf/{name}/x"], g[f"tf/{name}/y"].float(), g[f"tf/{name}/args"]
        zoom, cubic, padded = float(args[0]), bool(args[1]), tuple(int(v) for v in args[2:])
        dst = torch.full(padded, 9.0, device=DEV)
        K.zoom_scale_pad(x.to(DEV), (zoom,) * x.dim(), dst, cubic=cubic)
        assert float((dst.cpu() - want).abs().max()) <= 2e-5, (name, float((dst.cpu() - want).abs().max()))  # fp32 interpolation weights vs float64
