#!/usr/bin/env python
"""
Generate tests/golden/*.npz.  TEST INFRASTRUCTURE ONLY.

Part 1 (``hdr_ops_reference.npz``) runs the REFERENCE's own
``gm_diffusion/stage1/tone_mapping.py`` and ``augmentations.py`` -- loaded by
file path, because ``import gm_diffusion`` needs ``diffusers`` which is not
installed -- on seeded inputs and stores inputs + the reference's outputs.
This only works in the build container (/root/reference is mounted there); the
fixtures are committed so the tests never need the reference.

Part 2 (``pipeline_oracle_*.npz``) stores whole-pipeline vectors produced by the
oracle itself (seeded tiny random-weight UNets): they pin the product's host
logic / GPU path to the oracle, not the oracle to diffusers (parity unpinned).

Part 3 (``--slow``; ``pipeline_oracle_dual_sd15_512.npz``) is the north-star path at its own width -- the dual-UNet loop with
both SD-1.5-width UNets at 512x512, 50 PNDM steps -- from the same oracle; minutes of CPU, regenerated only on request.

Usage:  python oracle/make_golden.py [--reference /root/reference] [--slow]
"""
from __future__ import annotations

import argparse
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def make_hdr_reference(ref_root):
    tm = _load(os.path.join(ref_root, "gm_diffusion/stage1/tone_mapping.py"), "_ref_tone_mapping")
    aug = _load(os.path.join(ref_root, "gm_diffusion/stage1/augmentations.py"), "_ref_augmentations")

    g = torch.Generator().manual_seed(20251017)
    B, H, W = 2, 24, 40
    # decoder-like inputs in [-1.2, 1.2] so that the clamps are exercised
    sdr_dec = (torch.rand(B, 3, H, W, generator=g) * 2.4 - 1.2).float()
    gm_dec = (torch.rand(B, 3, H, W, generator=g) * 2.4 - 1.2).float()
    sdr = (sdr_dec / 2 + 0.5).clamp(0, 1)
    gm = (gm_dec / 2 + 0.5).clamp(0, 1)
    # edge values: exact 0, 1, values on .5/65535 and k/255 boundaries
    edge = torch.tensor(
        [0.0, 1.0, 0.5, 1 / 255, 2 / 255, 254 / 255, 255 / 256, 0.999999, 1e-8, 0.25,
         0.5 / 65535, 1.5 / 65535, 2.5 / 65535, 32767.5 / 65535, 65534.5 / 65535, -0.1, 1.1],
        dtype=torch.float32,
    )
    out = {
        "sdr_dec": sdr_dec.numpy(), "gm_dec": gm_dec.numpy(),
        "sdr": sdr.numpy(), "gm": gm.numpy(), "edge": edge.numpy(),
    }
    for qmax in (9, 49, 99):
        hdr = tm.apply_gm_to_sdr(gm, sdr, qmax=qmax)
        out[f"apply_gm_to_sdr_q{qmax}"] = hdr.numpy()
        out[f"fix_mulog_tmo_q{qmax}"] = tm.fix_mulog_tmo(hdr, qmax).numpy()
        out[f"linear_scale_tmo_q{qmax}"] = tm.linear_scale_tmo(hdr, qmax).numpy()
        out[f"hard_clip_tmo_q{qmax}"] = tm.hard_clip_tmo(hdr, qmax).numpy()
        out[f"stage1_chain_q{qmax}"] = tm.gamut_compress(tm.fix_mulog_tmo(hdr, qmax)).numpy()
    out["apply_gm_to_sdr_default"] = tm.apply_gm_to_sdr(gm, sdr).numpy()
    hdr10 = tm.apply_gm_to_sdr(gm, sdr, qmax=9)
    out["tmo_cuda"] = tm.tmo_cuda(hdr10).numpy()
    out["gamut_compress"] = tm.gamut_compress(sdr).numpy()
    out["gamut_compress_hdr"] = tm.gamut_compress(hdr10).numpy()
    # random_tmo_cuda with the python RNG pinned; record the mu it drew
    import random

    random.seed(7)
    mu = random.uniform(500, 5_000)
    random.seed(7)
    out["random_tmo_cuda_q49"] = tm.random_tmo_cuda(tm.apply_gm_to_sdr(gm, sdr, qmax=49), 49).numpy()
    out["random_tmo_mu"] = np.float64(mu)
    # uint16 discretisation (aug.py:38-41)
    x16 = torch.cat([torch.rand(4096, generator=g) * 1.2 - 0.1, edge]).float()
    out["u16_in"] = x16.numpy()
    out["discretize_to_uint16"] = aug.RandomExposureAdjust.discretize_to_uint16(x16).numpy()
    out["edge_apply_q99"] = tm.apply_gm_to_sdr(edge.flip(0).clamp(0, 1), edge, qmax=99).numpy()
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, "hdr_ops_reference.npz"), **out)
    print("wrote hdr_ops_reference.npz with", len(out), "arrays")


def make_pipeline_vectors(slow=False):
    sys.path.insert(0, ROOT)
    from oracle import fixtures

    os.makedirs(GOLD, exist_ok=True)
    todo = fixtures.SLOW_PIPELINE_FIXTURES if slow else fixtures.PIPELINE_FIXTURES
    for name, fn in todo.items():
        out = fn()
        np.savez_compressed(os.path.join(GOLD, f"pipeline_oracle_{name}.npz"), **out)
        print("wrote pipeline_oracle_%s.npz" % name, {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--skip-pipeline", action="store_true")
    ap.add_argument("--slow", action="store_true",
                    help="only the full-width fixtures (pipeline_oracle_dual_sd15_512.npz: ~150 SD-1.5 UNet evaluations on the CPU, minutes)")
    a = ap.parse_args()
    if a.slow:
        import time

        torch.set_num_threads(os.cpu_count() or 1)
        t0 = time.time()
        make_pipeline_vectors(slow=True)
        print("slow fixtures: %.0f s on %d threads" % (time.time() - t0, torch.get_num_threads()))
        sys.exit(0)
    if os.path.isdir(a.reference):
        make_hdr_reference(a.reference)
    else:
        print("reference not present; hdr_ops_reference.npz not regenerated")
    if not a.skip_pipeline:
        make_pipeline_vectors()
