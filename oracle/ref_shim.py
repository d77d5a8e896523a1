"""Import harness for the upstream reference (TEST INFRASTRUCTURE, never shipped, never on the product path).

Only usable where ``/root/reference`` exists (the build container). It installs three small stand-in
modules so that the reference's *model* files import without their heavy optional dependencies:

* ``cinema`` / ``cinema.mae`` / ``cinema.segmentation`` namespace packages pointing into
  ``/root/reference/cinema`` (bypasses ``cinema/__init__.py`` which pulls in monai),
* ``timm`` -- the handful of layers the reference uses, restated from timm 1.0.15 semantics
  (``Mlp`` = fc1 -> act -> drop -> norm -> fc2 -> drop; ``use_conv`` swaps in 1x1 ``Conv2d``),
* ``omegaconf`` -- attribute-access dict config backed by PyYAML.

Nothing here copies reference source; the reference is imported from where it lies.
"""

from __future__ import annotations

import sys
import types
from functools import partial
from itertools import repeat
from pathlib import Path

REFERENCE_ROOT = Path("/root/reference")


def reference_available() -> bool:
    return (REFERENCE_ROOT / "cinema" / "mae" / "mae.py").exists()


def _make_timm() -> None:
    import torch
    from torch import nn

    def to_2tuple(x):  # noqa: ANN001, ANN202
        if isinstance(x, (tuple, list)):
            return tuple(x)
        return tuple(repeat(x, 2))

    class DropPath(nn.Module):
        def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True) -> None:
            super().__init__()
            self.drop_prob = drop_prob
            self.scale_by_keep = scale_by_keep

        def forward(self, x):  # noqa: ANN001, ANN202
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1.0 - self.drop_prob
            shape = (x.shape[0],) + (1,) * (x.ndim - 1)
            noise = x.new_empty(shape).bernoulli_(keep)
            if keep > 0.0 and self.scale_by_keep:
                noise.div_(keep)
            return x * noise

    class Mlp(nn.Module):
        def __init__(  # noqa: PLR0913
            self,
            in_features,  # noqa: ANN001
            hidden_features=None,  # noqa: ANN001
            out_features=None,  # noqa: ANN001
            act_layer=nn.GELU,  # noqa: ANN001
            norm_layer=None,  # noqa: ANN001
            bias=True,  # noqa: ANN001, FBT002
            drop=0.0,  # noqa: ANN001
            use_conv=False,  # noqa: ANN001, FBT002
        ) -> None:
            super().__init__()
            out_features = out_features or in_features
            hidden_features = hidden_features or in_features
            bias = to_2tuple(bias)
            drop = to_2tuple(drop)
            layer = partial(nn.Conv2d, kernel_size=1) if use_conv else nn.Linear
            self.fc1 = layer(in_features, hidden_features, bias=bias[0])
            self.act = act_layer()
            self.drop1 = nn.Dropout(drop[0])
            self.norm = norm_layer(hidden_features) if norm_layer is not None else nn.Identity()
            self.fc2 = layer(hidden_features, out_features, bias=bias[1])
            self.drop2 = nn.Dropout(drop[1])

        def forward(self, x):  # noqa: ANN001, ANN202
            return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))

    class SwiGLU(nn.Module):
        """Placeholder: the hot path never instantiates it (only compared by identity)."""

    class LayerScale(nn.Module):
        def __init__(self, dim: int, init_values: float = 1e-5, inplace: bool = False) -> None:
            super().__init__()
            self.inplace = inplace
            self.gamma = nn.Parameter(init_values * torch.ones(dim))

        def forward(self, x):  # noqa: ANN001, ANN202
            return x.mul_(self.gamma) if self.inplace else x * self.gamma

    def param_groups_weight_decay(model, weight_decay=1e-5, no_weight_decay_list=()):  # noqa: ANN001, ANN202
        skip = set(no_weight_decay_list)
        decay, no_decay = [], []
        for name, p in model.named_parameters():
            if not p.requires_grad:
                continue
            (no_decay if (p.ndim <= 1 or name.endswith(".bias") or name in skip) else decay).append(p)
        return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]

    timm = types.ModuleType("timm")
    layers = types.ModuleType("timm.layers")
    models = types.ModuleType("timm.models")
    vt = types.ModuleType("timm.models.vision_transformer")
    optim = types.ModuleType("timm.optim")
    layers.DropPath, layers.Mlp, layers.SwiGLU, layers.to_2tuple = DropPath, Mlp, SwiGLU, to_2tuple
    layers.use_fused_attn = lambda: False  # explicit matmul-softmax branch: plain fp32 math on CPU
    vt.LayerScale = LayerScale
    optim.param_groups_weight_decay = param_groups_weight_decay
    timm.layers, timm.models, timm.optim, models.vision_transformer = layers, models, optim, vt
    for name, mod in {
        "timm": timm,
        "timm.layers": layers,
        "timm.models": models,
        "timm.models.vision_transformer": vt,
        "timm.optim": optim,
    }.items():
        sys.modules[name] = mod


class _Cfg(dict):
    """Attribute-access dict, enough for ``get_model(config)``."""

    def __getattr__(self, k):  # noqa: ANN001, ANN204
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v) -> None:  # noqa: ANN001
        self[k] = v


def _wrap(obj):  # noqa: ANN001, ANN202
    if isinstance(obj, dict):
        return _Cfg({k: _wrap(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [_wrap(v) for v in obj]
    return obj


def _make_omegaconf() -> None:
    import yaml

    class OmegaConf:
        @staticmethod
        def create(obj=None):  # noqa: ANN001, ANN205
            return _wrap(obj or {})

        @staticmethod
        def load(path):  # noqa: ANN001, ANN205
            with open(path, encoding="utf-8") as f:
                return _wrap(yaml.safe_load(f))

        @staticmethod
        def save(config, f):  # noqa: ANN001, ANN205
            with open(f, "w", encoding="utf-8") as fh:
                yaml.safe_dump(config, fh)

        @staticmethod
        def to_container(cfg, resolve=True):  # noqa: ANN001, ANN205, ARG004, FBT002
            return cfg

    m = types.ModuleType("omegaconf")
    m.DictConfig, m.OmegaConf = _Cfg, OmegaConf
    sys.modules["omegaconf"] = m


def install() -> None:
    """Make ``from cinema.mae.mae import CineMA`` resolve to the upstream reference."""
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True  # /root/reference is read-only
    if "timm" not in sys.modules:
        _make_timm()
    if "omegaconf" not in sys.modules:
        _make_omegaconf()
    for name, sub in (("cinema", ""), ("cinema.mae", "mae"), ("cinema.segmentation", "segmentation")):
        if name in sys.modules and getattr(sys.modules[name], "__ref_shim__", False):
            continue
        pkg = types.ModuleType(name)
        pkg.__path__ = [str(REFERENCE_ROOT / "cinema" / sub)] if sub else [str(REFERENCE_ROOT / "cinema")]
        pkg.__ref_shim__ = True
        sys.modules[name] = pkg


def cfg(obj: dict) -> _Cfg:
    return _wrap(obj)
