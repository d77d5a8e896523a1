"""GPU input pipeline for the pre-training step (reference ``cinema/mae/pretrain.py:157-200`` runs monai ``RandZoomd`` ->
``ScaleIntensityd`` -> ``SpatialPadd(method="end")`` in CPU DataLoader workers).

Here the raw fp32 samples travel host -> device through pinned staging buffers on a copy stream (double-buffered: the upload of batch i+1 overlaps
the training step of batch i) and the three transforms are two HIP launches per image, written straight into the padded batch tensors the model reads.
The random decisions (apply with probability ``prob``; one zoom factor in [min_zoom, max_zoom] shared by all axes, the SAX volume on its own and the
three long-axis views together, like the two ``RandZoomd`` entries of the reference) are drawn on the host from a seeded generator.

Parity note: monai is not installed in the build container and the reference holds no value test for these transforms, so monai itself cannot be run
against them: the kernels are checked against a torch restatement of monai 1.5.2's ``Zoom(keep_size=True)`` / ``ScaleIntensity`` / ``SpatialPad`` (oracle
``input_transform``), which is pinned against an independent second statement of the same published algorithm (float64 loops and explicit index maps) on
committed vectors (``tests/golden/second_opinion.safetensors``, ``tests/test_data_gpu.py::test_zoom_scale_pad_vs_the_pinned_second_opinion_vectors``).
"""

from __future__ import annotations

import numpy as np
import torch

from cinema_amd import hip as K

LAX_VIEWS = ("lax_2c", "lax_3c", "lax_4c")


class GpuInputPipeline:
    """``submit(samples)`` starts the upload of a list of per-subject dicts {view: fp32 host tensor (*size)}; ``get()`` returns the transformed
    batch {view: (batch, 1, *padded_size)} on the device (stream-ordered on the current stream)."""

    def __init__(self, padded_size_dict: dict, device: torch.device | str = "cuda", prob: float = 0.5, min_zoom: float = 0.9, max_zoom: float = 1.1,
                 seed: int = 0) -> None:
        self.sizes = {v: tuple(s) for v, s in padded_size_dict.items()}
        self.device = torch.device(device)
        self.prob, self.min_zoom, self.max_zoom = prob, min_zoom, max_zoom
        self.rng = np.random.default_rng(seed)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._slots: list = [None, None]   # double buffer: (pinned host tensors, device tensors, event, zooms)
        self._next, self._ready = 0, []

    def _draw(self) -> float:
        return float(self.rng.uniform(self.min_zoom, self.max_zoom)) if self.rng.random() < self.prob else 1.0

    def submit(self, samples: list) -> None:
        slot = self._next
        self._next ^= 1
        host, dev, zooms = [], [], []
        with torch.cuda.stream(self.copy_stream):
            for sample in samples:
                z_sax, z_lax = self._draw(), self._draw()  # RandZoomd(keys="sax") and RandZoomd(keys=lax views): one factor each per subject
                h, d = {}, {}
                for v, x in sample.items():
                    x = x.float().contiguous()
                    if any(a > b for a, b in zip(x.shape, self.sizes[v])):
                        raise ValueError(f"view {v}: sample size {tuple(x.shape)} exceeds the padded size {self.sizes[v]}")
                    h[v] = x if x.is_pinned() else x.pin_memory()
                    d[v] = h[v].to(self.device, non_blocking=True)
                host.append(h)
                dev.append(d)
                zooms.append({v: (z_sax if v == "sax" else z_lax) for v in sample})
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self._slots[slot] = (host, dev, ev, zooms)
        self._ready.append(slot)

    def get(self) -> dict:
        if not self._ready:
            raise RuntimeError("get() without a submitted batch")
        host, dev, ev, zooms = self._slots[self._ready.pop(0)]
        torch.cuda.current_stream().wait_event(ev)
        views = list(dev[0].keys())
        out = {v: torch.empty((len(dev), 1, *self.sizes[v]), dtype=torch.float32, device=self.device) for v in views}
        consumer = torch.cuda.current_stream()
        for i, (d, z) in enumerate(zip(dev, zooms)):
            for v in views:
                nd = d[v].dim()
                K.zoom_scale_pad(d[v], (z[v],) * nd, out[v][i, 0], cubic=(nd == 2))  # trilinear SAX, bicubic LAX (pretrain.py:170,178)
                # the raw sample was allocated under the copy stream but is read by the zoom kernel on the compute stream: without this the block goes
                # back to the copy stream's pool when the slot is overwritten and the next upload may land on it while the zoom is still queued
                d[v].record_stream(consumer)
        del host
        return out
