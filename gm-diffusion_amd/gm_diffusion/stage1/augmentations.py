"""
``RandomExposureAdjust`` -- kept as an exported NAME (reference ``gm_diffusion/stage1/__init__.py:18-28``)
for the one member that sits on the Stage-3 hot path: the uint16 gain-map quantiser
``discretize_to_uint16`` (reference gm_diffusion/stage1/augmentations.py:38-41; the "uint16
gain-map quantisation" of the north star), which runs as a HIP kernel, bit-exact.

The rest of the reference class (random exposure levels, camera-curve sampling, the HDR->LDR
re-exposure in ``__call__``) is Stage-1 *training-time* data augmentation: SURVEY.md §2 row 6 marks
it out of scope for this build, so it is deliberately not provided -- those members raise
``OutOfScopeError`` instead of computing on the host.
"""
from __future__ import annotations

import torch

from .. import hip_ops as ops


class OutOfScopeError(NotImplementedError, AttributeError):
    """A reference entry point outside the Stage-3 hot path was called.  Also an ``AttributeError``, so
    ``hasattr(obj, "gamma")`` answers False instead of raising."""


def _as_f32(img: torch.Tensor) -> torch.Tensor:
    img = img.contiguous()
    return img if img.dtype == torch.float32 else ops.cast(img, torch.float32)


class RandomExposureAdjust:
    _OUT_OF_SCOPE = ("Stage-1 training-time augmentation is out of scope for the MI355X hot-path build (SURVEY.md §2 row 6); "
                     "only RandomExposureAdjust.discretize_to_uint16 / uint16_codes are provided")

    def __init__(self, *args, **kwargs):
        # constructing the object is harmless (callers reach the static quantiser through an instance too)
        self._ctor_args = (args, kwargs)

    @staticmethod
    def discretize_to_uint16(img: torch.Tensor) -> torch.Tensor:
        """float image in [0, 1] -> the nearest of the 65,536 uint16 levels, returned as float32
        (``round`` = round-half-to-even, like torch).  Device tensors only: one HIP kernel."""
        return ops.discretize_u16(_as_f32(img))

    @staticmethod
    def uint16_codes(img: torch.Tensor) -> torch.Tensor:
        """The integer codes behind :meth:`discretize_to_uint16` (uint16 tensor, same shape)."""
        return ops.discretize_u16(_as_f32(img), codes=True)[1]

    # -- the reference's Stage-1 members (augmentations.py:10-80): named explicitly so that class-level access
    # (``RandomExposureAdjust.sample_camera_curve()``) and instance access fail the same, loud way ------------------------
    def __call__(self, *args, **kwargs):
        raise OutOfScopeError(self._OUT_OF_SCOPE)

    @staticmethod
    def hdr_to_ldr(*args, **kwargs):
        raise OutOfScopeError(RandomExposureAdjust._OUT_OF_SCOPE)

    @staticmethod
    def sample_camera_curve(*args, **kwargs):
        raise OutOfScopeError(RandomExposureAdjust._OUT_OF_SCOPE)

    @staticmethod
    def apply_inv_sigmoid_curve(*args, **kwargs):
        raise OutOfScopeError(RandomExposureAdjust._OUT_OF_SCOPE)

    def _out_of_scope_attr(self):
        raise OutOfScopeError(self._OUT_OF_SCOPE)

    exposure_levels = property(_out_of_scope_attr)
    gamma = property(_out_of_scope_attr)
    prob = property(_out_of_scope_attr)

    def __repr__(self) -> str:  # pragma: no cover
        return "RandomExposureAdjust(<hot-path subset: discretize_to_uint16, uint16_codes>)"
