"""
Stage 1 utilities on the Stage-3 path: tone-mapping operators and the uint16 discretiser
(export list of the reference's gm_diffusion/stage1/__init__.py:6-28 minus ``Discriminator``,
the Stage-1 GAN critic, which is training-only and out of scope: SURVEY.md §2 row 9).
"""

from .augmentations import RandomExposureAdjust
from .tone_mapping import (
    apply_gm_to_sdr,
    fix_mulog_tmo,
    gamut_compress,
    hard_clip_tmo,
    linear_scale_tmo,
    random_tmo_cuda,
    tmo_cuda,
)

__all__ = [
    "RandomExposureAdjust",
    "apply_gm_to_sdr",
    "fix_mulog_tmo",
    "gamut_compress",
    "hard_clip_tmo",
    "linear_scale_tmo",
    "random_tmo_cuda",
    "tmo_cuda",
]
