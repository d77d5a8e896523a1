"""Rotary position embedding (interface of the reference ``cinema/rotary.py:12-128``).

``rotate_half`` / ``apply_rotary_emb`` / ``RotaryEmbedding`` keep the reference's names, signatures and results for callers that use them
directly; they are written from the kernel's own formula (pairwise rotation of the two feature halves), not from the reference's tiled tables.  Inside ``Attention(rotary=True)`` the rotation is the HIP kernel
``cinema_rope_heads`` applied in place to the fused q|k projection (``cinema_amd.tape.op_self_attention``): the reference hands the
module q, k of shape (batch, heads, tokens, head_dim) (``cinema/vit.py:496-499``), so the table is indexed by the HEAD (dim 1) and the
angle is the same for every token - :meth:`RotaryEmbedding.head_tables` builds exactly that (heads, head_dim/2) table.
"""

from __future__ import annotations

import torch


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    """(..., d) -> (-second half, first half) of the last axis (known answer: ``cinema/rotary_test.py:9-13``)."""
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rotary_emb(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """The rotation ``cinema_rope_heads`` performs, as a tensor function with the reference's calling convention (``cinema/rotary.py:27-60``):
    x (batch, n_x_tokens, n_heads, head_dim), cos / sin (n_tokens >= n_x_tokens, r) with 2r <= head_dim.  Feature pairs (j, j + r), j < r, of token t
    turn by the angle of table row t: (a, b) -> (a cos - b sin, b cos + a sin); features from 2r on pass through."""
    r = cos.shape[-1]
    if 2 * r > x.shape[-1]:
        raise ValueError(f"Rotary dim {2 * r} is larger than the last dimension of x {x.shape[-1]}")
    n = x.shape[1]
    c, s = cos[:n].unsqueeze(1), sin[:n].unsqueeze(1)  # one row per token, shared by the heads
    a, b, tail = x[..., :r], x[..., r:2 * r], x[..., 2 * r:]
    return torch.cat((a * c - b * s, b * c + a * s, tail), dim=-1)


class RotaryEmbedding(torch.nn.Module):
    """Angle tables theta[t, j] = (t / scaling_factor) * base^(-2j / dim) and the rotation of (q, k) by them; constructor and call signature of the
    reference module (``cinema/rotary.py:63-128``).  ``inv_freq`` is a NON-persistent buffer: the module adds nothing to a ``state_dict``.  The tables
    are cached per (device, dtype) and grown on demand; :meth:`head_tables` is what the HIP attention path uses."""

    def __init__(self, dim: int, base: float = 10000.0, scaling_factor: float = 1.0, device: torch.device | None = None) -> None:
        super().__init__()
        self.dim, self.base, self.scaling_factor = dim, float(base), scaling_factor
        exponent = torch.arange(0, dim, 2, device=device, dtype=torch.float32) / dim
        self.register_buffer("inv_freq", self.base ** -exponent, persistent=False)
        self._tables: dict = {}
        self._head_tables: dict = {}

    def _angles(self, n: int, device: torch.device | str) -> torch.Tensor:
        t = torch.arange(n, dtype=torch.float32, device=device) / self.scaling_factor
        return t[:, None] * self.inv_freq.detach().to(device=device, dtype=torch.float32)[None, :]

    def tables(self, n_tokens: int, device: torch.device, dtype: torch.dtype) -> tuple:
        """(cos, sin), each (>= n_tokens, dim / 2) in ``dtype`` on ``device``."""
        key = (str(device), dtype)
        hit = self._tables.get(key)
        stale = hit is not None and self.training and hit[0].is_inference()  # tables built under inference_mode cannot feed autograd
        if hit is None or hit[0].shape[0] < n_tokens or stale:
            ang = self._angles(n_tokens, device)
            hit = self._tables[key] = (ang.cos().to(dtype), ang.sin().to(dtype))
        return hit

    def forward(self, q: torch.Tensor, k: torch.Tensor, offset: int = 0) -> tuple:
        if q.shape[1] != k.shape[1]:
            raise ValueError("q and k must have the same sequence length")
        cos, sin = self.tables(q.shape[1] + offset, q.device, q.dtype)
        return apply_rotary_emb(q, cos[offset:], sin[offset:]), apply_rotary_emb(k, cos[offset:], sin[offset:])

    def head_tables(self, n_heads: int, device: torch.device) -> tuple:
        """fp32 (n_heads, dim/2) cos / sin tables of the call ``self(q, k)`` with q, k (batch, n_heads, tokens, head_dim): the module's
        "token" axis is the head axis there.  Shape-only constants, cached per (heads, device)."""
        key = (n_heads, str(device))
        hit = self._head_tables.get(key)
        if hit is None:
            ang = self._angles(n_heads, "cpu")
            hit = self._head_tables[key] = (ang.cos().contiguous().to(device), ang.sin().contiguous().to(device))
        return hit
