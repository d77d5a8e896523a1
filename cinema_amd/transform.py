"""Sliding-window bookkeeping of the evaluation path (interface of the reference ``cinema/transform.py:13-143``): host-side index arithmetic
and tensor slicing on the caller's device.  The arithmetic of the segmentation sliding window itself (softmax, overlap averaging, log) runs in
the HIP kernels behind ``cinema_amd.segmentation.train.segmentation_forward``."""

from __future__ import annotations

import numpy as np
import torch


def get_patch_grid(image_size: tuple, patch_size: tuple, patch_overlap: tuple) -> np.ndarray:
    """Window start indices (n_patches, n) on a regular grid whose last window is shifted back to end at the border
    (reference ``cinema/transform.py:13-50``; known answers ``cinema/transform_test.py:13-96``)."""
    indices = []
    for size, patch, overlap in zip(image_size, patch_size, patch_overlap):
        if patch > size:
            raise ValueError(f"Patch size {patch} should be <= image size {size}.")
        end = size - patch + 1
        starts = np.arange(0, end, patch - overlap)
        if starts[-1] != end - 1:
            starts = np.append(starts, size - patch)
        indices.append(starts)
    return np.stack(np.meshgrid(*indices, indexing="ij"), axis=-1).reshape(-1, len(image_size))


def patch_grid_sample(x: torch.Tensor, start_indices: np.ndarray, patch_size: tuple) -> torch.Tensor:
    """Stack of windows: (n_patches, *patch) from (*size) or (n_patches, ch, *patch) from (ch, *size) (reference ``transform.py:53-83``)."""
    n = len(patch_size)
    lead = () if x.ndim == n else (slice(None),)
    return torch.stack([x[lead + tuple(slice(int(s[i]), int(s[i]) + patch_size[i]) for i in range(n))] for s in start_indices])


def aggregate_patches(patches: torch.Tensor, start_indices: np.ndarray, image_size: tuple) -> torch.Tensor:
    """(n_patches, ch, *patch) -> (ch, *image_size): mean over the windows covering each position (reference ``transform.py:86-124``)."""
    n_patches, ch, *patch_size = patches.shape
    n_dims = len(image_size)
    if n_patches != start_indices.shape[0]:
        raise ValueError(f"n_patches should be the same as start_indices, got {n_patches} and {start_indices.shape[0]}.")
    if n_dims != len(patch_size):
        raise ValueError(f"image_size and patch_size should have the same length, got image_size={image_size} and patches.shape={patches.shape}.")
    x = torch.zeros((ch, *image_size), dtype=patches.dtype, device=patches.device)
    count = torch.zeros(tuple(image_size), dtype=torch.float32, device=patches.device)
    for i in range(n_patches):
        sl = tuple(slice(int(start_indices[i][d]), int(start_indices[i][d]) + patch_size[d]) for d in range(n_dims))
        x[(slice(None), *sl)] += patches[i]
        count[sl] += 1
    return x / count[None, ...]


def crop_start(image, target_shape: tuple):  # noqa: ANN001, ANN201
    """image[:s0, :s1, ...] (reference ``transform.py:127-143``; padding is added at the end, so the crop keeps the start)."""
    if len(image.shape) != len(target_shape):
        raise ValueError(f"image.shape and target_shape should have the same length, got {image.shape} and {target_shape}.")
    return image[tuple(slice(0, s) for s in target_shape)]
