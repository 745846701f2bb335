"""
Host-side helpers the reference pipelines take from diffusers: ``randn_tensor``
(stable_diffusion_gm.py:709-710), ``VaeImageProcessor.postprocess`` (:1106),
``StableDiffusionPipelineOutput`` (:1114).
"""
from __future__ import annotations

from dataclasses import dataclass, fields
from typing import Any, List, Optional, Union

import numpy as np
import torch

from .. import hip_ops as ops


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    """diffusers.utils.torch_utils.randn_tensor: draw on the generator's device (CPU generators give
    device-independent, shard-independent noise), then move.  A list of generators draws one
    sample per generator."""
    device = torch.device(device) if device is not None else torch.device("cpu")
    batch = shape[0]
    layout = layout or torch.strided
    rand_device = device
    if generator is not None:
        gen_dev = generator.device.type if not isinstance(generator, list) else generator[0].device.type
        if gen_dev != device.type and gen_dev == "cpu":
            rand_device = torch.device("cpu")
        elif gen_dev != device.type and gen_dev == "cuda":
            raise ValueError(f"Cannot generate a {device} tensor from a generator of type {gen_dev}.")
    if isinstance(generator, list) and len(generator) == 1:
        generator = generator[0]
    if isinstance(generator, list):
        sub = (1,) + tuple(shape[1:])
        lat = [torch.randn(sub, generator=generator[i], device=rand_device, dtype=dtype, layout=layout) for i in range(batch)]
        return torch.cat(lat, dim=0).to(device)
    return torch.randn(tuple(shape), generator=generator, device=rand_device, dtype=dtype, layout=layout).to(device)


class _Output:
    """dataclass with tuple-style indexing (diffusers ``BaseOutput``)."""

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        return tuple(getattr(self, f.name) for f in fields(self) if getattr(self, f.name) is not None)[k]

    def to_tuple(self):
        return tuple(getattr(self, f.name) for f in fields(self))


@dataclass
class StableDiffusionPipelineOutput(_Output):
    images: Any
    nsfw_content_detected: Optional[List[bool]]


@dataclass
class UNetOutput(_Output):
    sample: torch.Tensor


class VaeImageProcessor:
    def __init__(self, vae_scale_factor=8, do_normalize=True):
        self.vae_scale_factor = vae_scale_factor
        self.config = type("cfg", (), {"vae_scale_factor": vae_scale_factor, "do_normalize": do_normalize})()

    @staticmethod
    def denormalize(images):
        """``(images / 2 + 0.5).clamp(0, 1)`` -- HIP kernel for device tensors."""
        if torch.is_tensor(images) and images.is_cuda:
            x = images.contiguous()
            if x.dtype != torch.float32:
                x = ops.cast(x, torch.float32)
            return ops.tmo(x, 4)
        return (images / 2 + 0.5).clamp(0, 1)

    @staticmethod
    def pt_to_numpy(images):
        return images.cpu().permute(0, 2, 3, 1).float().numpy()

    @staticmethod
    def numpy_to_pil(images):
        from PIL import Image

        if images.ndim == 3:
            images = images[None, ...]
        images = (images * 255).round().astype("uint8")
        if images.shape[-1] == 1:
            return [Image.fromarray(im.squeeze(), mode="L") for im in images]
        return [Image.fromarray(im) for im in images]

    def postprocess(self, image, output_type="pil", do_denormalize=None):
        if not torch.is_tensor(image):
            raise ValueError(f"Input for postprocessing is in incorrect format: {type(image)}. Only pytorch tensor is supported")
        if output_type not in ("latent", "pt", "np", "pil"):
            output_type = "np"
        if output_type == "latent":
            return image
        if do_denormalize is None:
            do_denormalize = [self.config.do_normalize] * image.shape[0]
        if all(do_denormalize):
            image = self.denormalize(image)
        elif any(do_denormalize):
            image = torch.stack([self.denormalize(image[i:i + 1])[0] if do_denormalize[i] else image[i] for i in range(image.shape[0])])
        if output_type == "pt":
            return image
        image = self.pt_to_numpy(image)
        if output_type == "np":
            return image
        return self.numpy_to_pil(image)
