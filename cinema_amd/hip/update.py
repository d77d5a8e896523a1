"""Optimiser kernels (csrc/optim.hip): gradient norm, clip coefficient, fused AdamW.

Part of the ctypes front of ``libcinema_hip.so`` (see ``cinema_amd/hip/__init__.py`` for the loader, the launch recorder, lane groups and the per-stream
workspaces); everything here is re-exported there, so callers keep writing ``hip.<name>``.  Module-level switches and the recorder state live in the package and
are read through it (``H.<NAME>``) so that an assignment ``hip.<NAME> = ...`` is seen by every family."""
from __future__ import annotations

import ctypes as C

import torch

from cinema_amd import hip as H
from cinema_amd.hip import (  # noqa: F401
    HipLibraryError, _check, _dev, _p, _stream, _workspace, load,
)

__all__ = ['ADAMW_MAX_GROUPS', 'AdamWGroup', 'adamw', 'adamw_groups', 'clip_coef', 'sqnorm']


def sqnorm(g: torch.Tensor, out: torch.Tensor) -> None:
    _dev(g, out)
    ws = _workspace("sqnorm", 2048, g.device)
    _check(load().cinema_sqnorm_f32(g.data_ptr(), g.numel(), out.data_ptr(), ws.data_ptr(), _stream()), "sqnorm")


def clip_coef(sq: torch.Tensor, max_norm: float, coef_out: torch.Tensor | None, norm_out: torch.Tensor | None, step_state: torch.Tensor | None = None) -> None:
    """coef = min(1, max_norm / (sqrt(sq) + 1e-6)); a non-finite norm gives coef = 0 (= skip, see :func:`adamw`).  ``step_state`` int32 [2]:
    [0] counts applied updates, [1] skipped ones."""
    _dev(sq, coef_out, norm_out, step_state)
    if step_state is not None and (step_state.dtype != torch.int32 or step_state.numel() < 2):
        raise HipLibraryError("clip_coef: step_state must be int32 [2]")
    _check(load().cinema_clip_coef(sq.data_ptr(), max_norm, _p(coef_out), _p(norm_out), _p(step_state), _stream()), "clip_coef")


def adamw(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float,
          step: int, clip: torch.Tensor | None = None, shadow: torch.Tensor | None = None, step_state: torch.Tensor | None = None) -> None:
    """``step_state`` (int32 [2] written by :func:`clip_coef`): the update is skipped on the device when ``clip[0]`` is not > 0 and the Adam
    step of the bias corrections is ``step_state[0]`` (``step`` is then ignored)."""
    _dev(p, g, m, v, clip, shadow, step_state)
    step = max(int(step), 1)
    _check(load().cinema_adamw(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps, weight_decay,
                               1.0 - beta1**step, 1.0 - beta2**step, _p(clip), _p(shadow), _p(step_state), _stream()), "adamw")


class AdamWGroup(C.Structure):
    """Mirror of ``cinema_adamw_group``."""

    _fields_ = [("begin", C.c_longlong), ("end", C.c_longlong), ("lr", C.c_float), ("weight_decay", C.c_float)]


ADAMW_MAX_GROUPS = 64


def adamw_groups(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, groups: list, beta1: float, beta2: float, eps: float,
                 clip: torch.Tensor, shadow: torch.Tensor | None, step_state: torch.Tensor, max_blocks: int = 0) -> None:
    """AdamW over several ranges of ONE flat buffer in one launch: ``groups`` = [(begin, end, lr, weight_decay), ...], ascending element ranges (multiples of 4).
    ``max_blocks`` caps the number of workgroups (0: the library's default) for an update that shares the chip with another stream's work.
    Same arithmetic per element as :func:`adamw` with ``step_state`` (the layer-decay groups of a fine-tuning step: one launch instead of one per group)."""
    _dev(p, g, m, v, clip, shadow, step_state)
    arr = (AdamWGroup * len(groups))()
    for a, (b, e, lr, wd) in zip(arr, groups):
        a.begin, a.end, a.lr, a.weight_decay = int(b), int(e), float(lr), float(wd)
    if max_blocks:
        _check(load().cinema_adamw_groups_grid(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), arr, len(groups), beta1, beta2, eps, clip.data_ptr(),
                                               _p(shadow), step_state.data_ptr(), int(max_blocks), _stream()), "adamw_groups_grid")
        return
    _check(load().cinema_adamw_groups(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), arr, len(groups), beta1, beta2, eps, clip.data_ptr(), _p(shadow),
                                      step_state.data_ptr(), _stream()), "adamw_groups")
