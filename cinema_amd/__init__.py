"""cinema_amd: MI355X-native (gfx950) compute path for the CineMA MAE hot path.

Public surface mirrors what callers of the reference import (``cinema/__init__.py:3-7,23-34``) for this path; the repo-root ``cinema``
package aliases these modules under the reference's import names.
"""

from cinema_amd.convvit import ConvViT
from cinema_amd.mae.mae import CineMA
from cinema_amd.segmentation.convunetr import ConvUNetR
from cinema_amd.vit import patchify, unpatchify

__all__ = ["CineMA", "ConvUNetR", "ConvViT", "patchify", "unpatchify"]
