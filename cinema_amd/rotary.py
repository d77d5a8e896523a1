"""Rotary position embedding (interface of the reference ``cinema/rotary.py:12-128``).

``rotate_half`` / ``apply_rotary_emb`` / ``RotaryEmbedding`` keep the reference's signatures and semantics for callers that use them
directly (plain tensor functions on the caller's device).  Inside ``Attention(rotary=True)`` the rotation is the HIP kernel
``cinema_rope_heads`` applied in place to the fused q|k projection (``cinema_amd.tape.op_self_attention``): the reference hands the
module q, k of shape (batch, heads, tokens, head_dim) (``cinema/vit.py:496-499``), so the table is indexed by the HEAD (dim 1) and the
angle is the same for every token - :meth:`RotaryEmbedding.head_tables` builds exactly that (heads, head_dim/2) table.
"""

from __future__ import annotations

import torch


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    """(..., d) -> cat(-x2, x1) over the two halves of the last axis (reference ``rotary.py:12-24``)."""
    x1, x2 = x.chunk(2, dim=-1)
    return torch.cat((-x2, x1), dim=-1)


def apply_rotary_emb(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x (batch, n_x_tokens, n_heads, head_dim); cos/sin (n_tokens, rotary_dim/2) (reference ``rotary.py:27-60``): the first
    ``rotary_dim`` features are rotated with the tables tiled ``[cos, cos]`` / ``[sin, sin]``, the rest pass through."""
    ro_dim = cos.shape[-1] * 2
    if ro_dim > x.shape[-1]:
        raise ValueError(f"Rotary dim {ro_dim} is larger than the last dimension of x {x.shape[-1]}")
    n_tokens = x.size(1)
    cos2 = torch.cat([cos[:n_tokens], cos[:n_tokens]], dim=-1)[:, None, :]  # einops "s d -> s 1 (2 d)"
    sin2 = torch.cat([sin[:n_tokens], sin[:n_tokens]], dim=-1)[:, None, :]
    return torch.cat([x[..., :ro_dim] * cos2 + rotate_half(x[..., :ro_dim]) * sin2, x[..., ro_dim:]], dim=-1)


class RotaryEmbedding(torch.nn.Module):
    """cos/sin cache keyed on (n_tokens, device, dtype, inference mode); ``inv_freq`` is a non-persistent buffer, so the module adds
    nothing to a ``state_dict`` (reference ``rotary.py:63-128``)."""

    def __init__(self, dim: int, base: float = 10000.0, scaling_factor: float = 1.0, device: torch.device | None = None) -> None:
        super().__init__()
        self.dim = dim
        self.base = float(base)
        self.scaling_factor = scaling_factor
        self.device = device
        self.n_tokens = 0
        self.cos = None
        self.sin = None
        inv_freq = 1 / (self.base ** (torch.arange(0, self.dim, 2, device=device, dtype=torch.float32) / self.dim))
        self.register_buffer("inv_freq", inv_freq, persistent=False)
        self._head_tables: dict = {}

    def update_cos_sin(self, n_tokens: int, device: torch.device, dtype: torch.dtype) -> None:
        if ((n_tokens > self.n_tokens) or (self.cos is None) or (self.cos.device != device) or (self.cos.dtype != dtype)
                or (self.training and self.cos.is_inference())):
            self.n_tokens = n_tokens
            t = torch.arange(n_tokens, device=device, dtype=self.inv_freq.dtype) / self.scaling_factor
            freqs = torch.outer(t, self.inv_freq.to(device))
            self.cos = torch.cos(freqs).to(dtype)
            self.sin = torch.sin(freqs).to(dtype)

    def forward(self, q: torch.Tensor, k: torch.Tensor, offset: int = 0) -> tuple:
        if q.shape[1] != k.shape[1]:
            raise ValueError("q and k must have the same sequence length")
        self.update_cos_sin(q.shape[1] + offset, device=q.device, dtype=q.dtype)
        return apply_rotary_emb(q, self.cos[offset:], self.sin[offset:]), apply_rotary_emb(k, self.cos[offset:], self.sin[offset:])

    def head_tables(self, n_heads: int, device: torch.device) -> tuple:
        """fp32 (n_heads, dim/2) cos / sin tables of the call ``self(q, k)`` with q, k (batch, n_heads, tokens, head_dim): the module's
        "token" axis is the head axis there.  Shape-only constants, cached per (heads, device)."""
        key = (n_heads, str(device))
        hit = self._head_tables.get(key)
        if hit is None:
            t = torch.arange(n_heads, dtype=torch.float32) / self.scaling_factor
            freqs = torch.outer(t, self.inv_freq.detach().float().cpu())
            hit = self._head_tables[key] = (torch.cos(freqs).contiguous().to(device), torch.sin(freqs).contiguous().to(device))
        return hit
