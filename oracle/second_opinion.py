"""TEST INFRASTRUCTURE - an independent SECOND statement of the third-party arithmetic the hot path calls but the reference repository holds no value for.

monai 1.5.2 (reference ``pyproject.toml``) is not installed in the build image and cannot be imported, so ``oracle/cinema_oracle.py`` restates
  * ``monai.losses.DiceLoss(include_background=False, to_onehot_y=False, softmax=True)`` + ``monai.networks.utils.one_hot`` as called at
    ``cinema/segmentation/train.py:93-100``,
  * ``monai.transforms.Zoom(keep_size=True, padding_mode="constant")`` (the deterministic core of ``RandZoomd``), ``ScaleIntensity()`` and
    ``SpatialPad(method="end")`` as composed at ``cinema/mae/pretrain.py:166-199``
with torch tensor ops.  This file states the same published algorithms a second time in a deliberately different style - float64 numpy, explicit loops over
samples / classes / voxels and explicit index maps, no torch.nn.functional - so that ``tests/test_oracle_golden.py`` can pin the oracle's restatement against
something that shares no code (and no vectorisation idiom) with it.  The vectors both agree on are committed as ``tests/golden/second_opinion.safetensors``
(written by ``oracle/make_golden_second_opinion.py``).

Published algorithms restated here:
  DiceLoss.forward      p = softmax(logits, channel axis); drop channel 0 of p and of the one-hot target; per (sample, class): I = sum_voxels p t,
                        G = sum t, P = sum p; f = 1 - (2 I + smooth_nr) / (G + P + smooth_dr) with smooth_nr = smooth_dr = 1e-5; reduction "mean" over
                        samples x classes.
  Zoom(keep_size=True)  output extent floor(n * zoom) per axis; resample with torch's interpolate rule for align_corners=False: source coordinate
                        x = (i + 0.5) * n_in / n_out - 0.5; linear: x clamped at 0, neighbours floor(x) and min(floor(x) + 1, n_in - 1); cubic: the
                        Keys kernel with a = -0.75 on floor(x) - 1 .. floor(x) + 2, indices clamped to the border; then centre pad (zeros, front = diff // 2) or
                        centre crop (start = diff // 2) back to the input extent.
  ScaleIntensity()      (x - min) / (max - min); an all-constant image becomes zeros (monai ``rescale_array``: ``arr * minv`` with minv = 0).
  SpatialPad("end")     zeros appended after the data on every axis up to the requested size.
Only ``tests/`` may import this module."""
from __future__ import annotations

import math

import numpy as np


# --------------------------------------------------------------------------------------------------------------- segmentation loss
def softmax_channels(logits: np.ndarray) -> np.ndarray:
    """(batch, c, *spatial) float64 -> probabilities, one voxel at a time."""
    b, c = logits.shape[:2]
    flat = logits.reshape(b, c, -1).astype(np.float64)
    out = np.empty_like(flat)
    for s in range(b):
        for v in range(flat.shape[2]):
            col = flat[s, :, v]
            m = col.max()
            e = np.array([math.exp(float(x - m)) for x in col])
            out[s, :, v] = e / e.sum()
    return out.reshape(logits.shape)


def dice_loss_no_background(logits: np.ndarray, labels: np.ndarray, smooth_nr: float = 1e-5, smooth_dr: float = 1e-5) -> float:
    """monai DiceLoss(include_background=False, softmax=True, squared_pred=False, jaccard=False, reduction="mean") on one-hot(labels.clamp(min=0)).
    ``logits`` (batch, c, *spatial), ``labels`` (batch, 1, *spatial) integers (-1 = ignored by the cross entropy, clamped to class 0 for the Dice target as the
    reference does at ``cinema/segmentation/train.py:93``)."""
    b, c = logits.shape[:2]
    p = softmax_channels(logits).reshape(b, c, -1)
    lab = labels.reshape(b, -1)
    terms = []
    for s in range(b):
        for k in range(1, c):                      # channel 0 (background) is excluded from prediction AND target
            inter = ground = pred = 0.0
            for v in range(lab.shape[1]):
                t = 1.0 if max(int(lab[s, v]), 0) == k else 0.0
                inter += p[s, k, v] * t
                ground += t
                pred += p[s, k, v]
            terms.append(1.0 - (2.0 * inter + smooth_nr) / (ground + pred + smooth_dr))
    return float(sum(terms) / len(terms))


def cross_entropy_ignore(logits: np.ndarray, labels: np.ndarray, ignore_index: int = -1) -> float:
    """Mean over the non-ignored voxels of -log softmax(logits)[label] (``F.cross_entropy(..., ignore_index=-1)``)."""
    b, c = logits.shape[:2]
    flat = logits.reshape(b, c, -1).astype(np.float64)
    lab = labels.reshape(b, -1)
    total, n = 0.0, 0
    for s in range(b):
        for v in range(lab.shape[1]):
            k = int(lab[s, v])
            if k == ignore_index:
                continue
            col = flat[s, :, v]
            m = col.max()
            lse = m + math.log(sum(math.exp(float(x - m)) for x in col))
            total += lse - float(col[k])
            n += 1
    return total / n


def segmentation_loss(logits: np.ndarray, labels: np.ndarray) -> dict:
    ce, dice = cross_entropy_ignore(logits, labels), dice_loss_no_background(logits, labels)
    return {"cross_entropy": ce, "mean_dice_loss": dice, "loss": ce + dice}


# --------------------------------------------------------------------------------------------------------------- loader transforms
def _linear_taps(i: int, n_in: int, n_out: int) -> list:
    x = max((i + 0.5) * n_in / n_out - 0.5, 0.0)
    i0 = int(math.floor(x))
    i0 = min(i0, n_in - 1)
    i1 = min(i0 + 1, n_in - 1)
    w = x - i0
    return [(i0, 1.0 - w), (i1, w)]


def _cubic_taps(i: int, n_in: int, n_out: int, a: float = -0.75) -> list:
    x = (i + 0.5) * n_in / n_out - 0.5
    i0 = int(math.floor(x))
    t = x - i0

    def k(d: float) -> float:  # Keys cubic convolution kernel
        d = abs(d)
        if d <= 1.0:
            return ((a + 2.0) * d - (a + 3.0)) * d * d + 1.0
        if d < 2.0:
            return ((a * d - 5.0 * a) * d + 8.0 * a) * d - 4.0 * a
        return 0.0

    return [(min(max(i0 + o, 0), n_in - 1), k(t - o)) for o in (-1, 0, 1, 2)]


def resample(x: np.ndarray, out_shape: tuple, cubic: bool) -> np.ndarray:
    """Separable resampling of a 2-D / 3-D array, one output voxel and one axis at a time."""
    cur = x.astype(np.float64)
    for ax, n_out in enumerate(out_shape):
        n_in = cur.shape[ax]
        nxt = np.zeros(cur.shape[:ax] + (n_out,) + cur.shape[ax + 1:], dtype=np.float64)
        for i in range(n_out):
            taps = _cubic_taps(i, n_in, n_out) if cubic else _linear_taps(i, n_in, n_out)
            acc = 0.0
            for j, w in taps:
                acc = acc + w * np.take(cur, j, axis=ax)
            idx = [slice(None)] * cur.ndim
            idx[ax] = i
            nxt[tuple(idx)] = acc
        cur = nxt
    return cur


def zoom_keep_size(x: np.ndarray, zoom: float, cubic: bool) -> np.ndarray:
    size = x.shape
    out_shape = tuple(int(math.floor(float(n) * zoom)) for n in size)
    y = resample(x, out_shape, cubic)
    out = np.zeros(size, dtype=np.float64)
    src, dst = [], []
    for n, o in zip(size, out_shape):
        diff = n - o
        half = abs(diff) // 2
        if diff >= 0:      # zoomed out: centre the smaller result, zeros around it
            src.append(slice(0, o)); dst.append(slice(half, half + o))
        else:              # zoomed in: centre crop
            src.append(slice(half, half + n)); dst.append(slice(0, n))
    out[tuple(dst)] = y[tuple(src)]
    return out


def scale_intensity(x: np.ndarray) -> np.ndarray:
    lo, hi = float(x.min()), float(x.max())
    if hi == lo:
        return np.zeros_like(x, dtype=np.float64)
    return (x.astype(np.float64) - lo) / (hi - lo)


def pad_end(x: np.ndarray, size: tuple) -> np.ndarray:
    out = np.zeros(tuple(max(s, n) for s, n in zip(size, x.shape)), dtype=np.float64)
    out[tuple(slice(0, n) for n in x.shape)] = x
    return out


def input_transform(x: np.ndarray, zoom: float, padded_size: tuple, cubic: bool) -> np.ndarray:
    """RandZoomd (when it fires, with this factor) -> ScaleIntensityd -> SpatialPadd(method="end") of one view (``cinema/mae/pretrain.py:166-199``)."""
    y = zoom_keep_size(x, zoom, cubic) if zoom != 1.0 else x.astype(np.float64)
    return pad_end(scale_intensity(y), padded_size)
