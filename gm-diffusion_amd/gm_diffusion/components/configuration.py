"""
Minimal stand-ins for the pieces of ``diffusers.configuration_utils`` the reference pipelines
touch: a frozen dict with attribute access (``scheduler.config.steps_offset``,
``dict(scheduler.config)``, ``X.from_config(pipeline.scheduler.config)`` --
stable_diffusion_gm.py:216-241, scripts/inference/experiments/formal_improved.py:195) and the
diffusers checkpoint directory layout (``config.json`` + ``diffusion_pytorch_model.safetensors``,
scripts/inference/generate_hdr.py:152-164).
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict


class FrozenDict(OrderedDict):
    """Read-only dict whose keys are also attributes (diffusers ``FrozenDict``)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__frozen = True

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name, value):
        if name.startswith("_"):
            return super().__setattr__(name, value)
        raise AttributeError(f"cannot set {name!r} on a frozen config")

    def __setitem__(self, name, value):
        if getattr(self, "_FrozenDict__frozen", False):
            raise TypeError("config is frozen; build a new FrozenDict")
        super().__setitem__(name, value)

    def __deepcopy__(self, memo):
        import copy

        return FrozenDict({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def __reduce__(self):
        return (FrozenDict, (dict(self),))


class ConfigMixin:
    """``self.config`` backed by ``self._internal_dict`` (the reference pipelines patch the latter directly:
    stable_diffusion_gm.py:226-228)."""

    config_name = "config.json"
    _defaults: dict = {}

    def register_to_config(self, **kwargs):
        cfg = dict(getattr(self, "_internal_dict", {}) or {})
        cfg.update(kwargs)
        self._internal_dict = FrozenDict(cfg)

    @property
    def config(self):
        return self._internal_dict

    @classmethod
    def load_config(cls, path, subfolder=None):
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, cls.config_name)) as f:
            return json.load(f)

    @classmethod
    def from_config(cls, config, **overrides):
        known = cls._defaults
        kw = {k: v for k, v in dict(config).items() if k in known}
        kw.update({k: v for k, v in overrides.items() if k in known})
        return cls(**kw)


def read_state_dict(directory, basename="diffusion_pytorch_model"):
    """Load a diffusers-layout weight file (safetensors preferred, .bin accepted) as CPU tensors."""
    st = os.path.join(directory, basename + ".safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file

        return load_file(st)
    b = os.path.join(directory, basename + ".bin")
    if os.path.exists(b):
        import torch

        return torch.load(b, map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"no {basename}.safetensors/.bin under {directory}")
