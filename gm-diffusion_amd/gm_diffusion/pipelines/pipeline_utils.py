"""
Minimal ``DiffusionPipeline`` for the MI355X build: the inherited surface the reference pipelines
and scripts use: ``register_modules``, ``from_pretrained(path, **component_overrides)``
(scripts/inference/generate_hdr.py:169-176), ``.to(device[, dtype])`` (:180), ``_execution_device``
(stable_diffusion_gm.py:960), ``progress_bar`` / ``set_progress_bar_config`` (:1039),
``maybe_free_model_hooks`` (:1109), ``enable_xformers_memory_efficient_attention``
(scripts/stage2/train_gm_unet.py:193-194).
"""
from __future__ import annotations

import importlib
import json
import os

import torch

from ..components import AutoencoderKL, CLIPTextModel, DDPMScheduler, DPMSolverMultistepScheduler, PNDMScheduler, UNet2DConditionModel
from ..components.configuration import FrozenDict

_LOADABLE = {
    "UNet2DConditionModel": UNet2DConditionModel,
    "AutoencoderKL": AutoencoderKL,
    "CLIPTextModel": CLIPTextModel,  # model_index.json lists it under "transformers": the HIP implementation takes over
    "PNDMScheduler": PNDMScheduler,
    "DDPMScheduler": DDPMScheduler,
    "DPMSolverMultistepScheduler": DPMSolverMultistepScheduler,
}


class _NullBar:
    def __init__(self, total=None, **kw):
        self.total, self.n = total, 0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def update(self, n=1):
        self.n += n


class DiffusionPipeline:
    config_name = "model_index.json"
    _optional_components = []

    def __init__(self):
        self._internal_dict = FrozenDict()
        self._component_names = []
        self._progress_bar_config = {}

    # ---- config ------------------------------------------------------------------------------
    @property
    def config(self):
        return self._internal_dict

    def register_to_config(self, **kwargs):
        d = dict(self._internal_dict)
        d.update(kwargs)
        self._internal_dict = FrozenDict(d)

    def register_modules(self, **kwargs):
        for name, module in kwargs.items():
            setattr(self, name, module)
            if name not in self._component_names:
                self._component_names.append(name)

    @property
    def components(self):
        return {k: getattr(self, k) for k in self._component_names}

    # ---- device ------------------------------------------------------------------------------
    def to(self, *args, **kwargs):
        device, dtype = kwargs.get("device"), kwargs.get("dtype") or kwargs.get("torch_dtype")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            elif a is not None:
                device = a
        for name in self._component_names:
            m = getattr(self, name)
            if m is None or not hasattr(m, "to") or name in ("scheduler", "tokenizer", "feature_extractor"):
                continue
            if isinstance(m, torch.nn.Module):
                m.to(device=device, dtype=dtype) if dtype is not None and any(p.is_floating_point() for p in m.parameters()) else m.to(device=device)
            else:
                m.to(device=device, dtype=dtype)
        return self

    @property
    def device(self):
        for name in self._component_names:
            m = getattr(self, name)
            d = getattr(m, "device", None)
            if isinstance(d, torch.device):
                return d
            if isinstance(m, torch.nn.Module):
                for p in m.parameters():
                    return p.device
        return torch.device("cpu")

    @property
    def _execution_device(self):
        return self.device

    # ---- misc --------------------------------------------------------------------------------
    def set_progress_bar_config(self, **kwargs):
        self._progress_bar_config = kwargs

    def progress_bar(self, iterable=None, total=None):
        cfg = dict(self._progress_bar_config)
        if cfg.get("disable", False):
            return _NullBar(total=total) if iterable is None else iterable
        try:
            from tqdm.auto import tqdm
        except Exception:  # pragma: no cover
            return _NullBar(total=total) if iterable is None else iterable
        return tqdm(iterable, **cfg) if iterable is not None else tqdm(total=total, **cfg)

    def maybe_free_model_hooks(self):
        pass

    def enable_xformers_memory_efficient_attention(self, *a, **k):
        """No-op: attention already runs in the LDS-tiled flash kernel (gmd_attention)."""

    def disable_xformers_memory_efficient_attention(self):
        pass

    # ---- loading -----------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, path, torch_dtype=None, **overrides):
        """Load a diffusers-layout pipeline directory; components passed as keyword arguments replace
        the ones on disk and are not loaded (generate_hdr.py:169-176)."""
        index = {}
        idx_path = os.path.join(path, cls.config_name)
        if os.path.exists(idx_path):
            with open(idx_path) as f:
                index = json.load(f)
        import inspect

        params = [p for p in inspect.signature(cls.__init__).parameters if p != "self"]
        kwargs = {}
        for name in params:
            if name in overrides:
                kwargs[name] = overrides[name]
                continue
            entry = index.get(name)
            if name == "requires_safety_checker":
                if name in index:
                    kwargs[name] = index[name]
                continue
            if not entry or entry[0] is None or name in ("safety_checker", "feature_extractor", "image_encoder"):
                if name in cls._optional_components or name in ("safety_checker", "feature_extractor", "image_encoder"):
                    kwargs[name] = None
                    continue
                raise ValueError(f"component {name!r} is neither in {idx_path} nor passed as an argument")
            lib_name, cls_name = entry
            sub = os.path.join(path, name)
            if cls_name in _LOADABLE:
                kwargs[name] = _LOADABLE[cls_name].from_pretrained(sub)
            elif lib_name == "transformers":
                kwargs[name] = getattr(importlib.import_module("transformers"), cls_name).from_pretrained(sub)
            elif cls_name.endswith("Scheduler"):
                # unknown scheduler class on disk: the SD-1.5 checkpoint ships PNDM
                kwargs[name] = PNDMScheduler.from_pretrained(sub)
            else:
                raise ValueError(f"cannot load component {name}: {lib_name}.{cls_name}")
        if "gm_unet" in params and "gm_unet" not in kwargs:
            raise ValueError("gm_unet must be passed explicitly (it is not part of the base checkpoint)")
        pipe = cls(**kwargs)
        if torch_dtype is not None:
            pipe.to(torch_dtype)
        return pipe
