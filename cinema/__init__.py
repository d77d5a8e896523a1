"""``cinema`` import surface of the reference (``cinema/__init__.py:3-34``) served by the MI355X build ``cinema_amd``.

``from cinema import CineMA, ConvViT, ConvUNetR, patchify, unpatchify`` and the sub-module imports the reference's training / inference
scripts use (``cinema.mae.mae``, ``cinema.convvit``, ``cinema.vit``, ``cinema.conv``, ``cinema.rotary``, ``cinema.optim``, ``cinema.device``, ``cinema.transform``,
``cinema.segmentation.convunetr``, ``cinema.segmentation.train``, ``cinema.classification.train``, ``cinema.regression.train``) resolve to the ``cinema_amd`` modules of the same name: one set of
classes, two import names.  Only the hot path is aliased; data loading, hydra entry points, the landmark models and the ResNet / UNet
baselines are not part of this build and are not faked here.
"""

import importlib
import sys

_ALIASES = {
    "cinema.vit": "cinema_amd.vit",
    "cinema.conv": "cinema_amd.conv",
    "cinema.convvit": "cinema_amd.convvit",
    "cinema.rotary": "cinema_amd.rotary",
    "cinema.optim": "cinema_amd.optim",
    "cinema.device": "cinema_amd.device",
    "cinema.transform": "cinema_amd.transform",
    "cinema.metric": "cinema_amd.metric",
    "cinema.mae": "cinema_amd.mae",
    "cinema.mae.mae": "cinema_amd.mae.mae",
    "cinema.mae.pretrain": "cinema_amd.mae.pretrain",
    "cinema.segmentation": "cinema_amd.segmentation",
    "cinema.segmentation.convunetr": "cinema_amd.segmentation.convunetr",
    "cinema.segmentation.train": "cinema_amd.segmentation.train",
    "cinema.train": "cinema_amd.train",
    "cinema.classification": "cinema_amd.classification",
    "cinema.classification.train": "cinema_amd.classification.train",
    "cinema.regression": "cinema_amd.regression",
    "cinema.regression.train": "cinema_amd.regression.train",
}
for _alias, _target in _ALIASES.items():
    _mod = importlib.import_module(_target)
    sys.modules[_alias] = _mod
    _parent, _, _leaf = _alias.rpartition(".")
    if _parent == "cinema":
        globals()[_leaf] = _mod

from cinema_amd import CineMA, ConvUNetR, ConvViT, patchify, unpatchify  # noqa: E402
from cinema_amd.metric import heatmap_soft_argmax  # noqa: E402

# dataset constants of the reference package root (cinema/__init__.py:9-21)
UKB_SPACING = (1.0, 1.0, 10.0)
UKB_LAX_SLICE_SIZE = (256, 256)
UKB_SAX_SLICE_SIZE = (192, 192)
UKB_N_FRAMES = 50

RV_LABEL = 1
MYO_LABEL = 2
LV_LABEL = 3
LABEL_TO_NAME = {RV_LABEL: "RV", MYO_LABEL: "MYO", LV_LABEL: "LV"}

__all__ = ["LABEL_TO_NAME", "LV_LABEL", "MYO_LABEL", "RV_LABEL", "CineMA", "ConvUNetR", "ConvViT", "heatmap_soft_argmax", "patchify", "unpatchify"]
