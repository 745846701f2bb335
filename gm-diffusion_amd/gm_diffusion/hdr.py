"""
The Stage-3 tail as ONE device-side pass: decode both latents, post-process, quantise the PNG
bytes, recompose HDR with Eq. 1 and scale for the Radiance writer.  This is the composition the
reference's CLI performs on the host after the pipeline returns latents
(scripts/inference/generate_hdr.py:225-265, scripts/inference/experiments/formal_improved.py:272-303):

    sdr = vae.decode(1/sf * sdr_latent); sdr = (sdr/2+0.5).clamp(0,1)        gen.py:225-228
    gm  = vae.decode(1/sf * gm_latent);  gm  = (gm/2+0.5).clamp(0,1)         gen.py:230-233
    PNG bytes (x*255).astype(uint8)                                          gen.py:244-245
    hdr = apply_gm_to_sdr(sdr, gm, qmax=99)   [numpy variant: no clamp]      fi.py:34-45, gen.py:256-259
    file = hdr/(qmax+1)                                                      gen.py:27-29

Here the two decodes run on the HIP VAE (channels-last) and everything after them is the single
``gmd_hdr_tail`` kernel reading the decoder's float32 [B,H*W,4] image directly (no transpose, no
host round trip).  Outputs stay on the device as [B,H,W,3] tensors; ``to_host`` gives the numpy
arrays the reference scripts hold.
"""
from __future__ import annotations

import torch

from . import hip_ops as ops


def decode_to_hdr(vae, sdr_latent, gm_latent, qmax=99.0, eps=1 / 64, clamp=False,
                  want=("sdr", "gm", "sdr_u8", "gm_u8", "hdr", "hdr_file", "hdr_u16")):
    """Returns a dict of device tensors [B,H,W,3]: sdr/gm (float32 in [0,1]), sdr_u8/gm_u8 (truncated PNG bytes),
    hdr (Eq. 1), hdr_file (= hdr/(qmax+1)), hdr_u16 (round-half-even codes of clamp(hdr_file))."""
    inv = 1.0 / vae.config.scaling_factor
    B = sdr_latent.shape[0]
    sdr_dec, H, W = vae.decode_nhwc(ops.tmo(_f32(sdr_latent), 5, mu=inv))
    gm_dec, _, _ = vae.decode_nhwc(ops.tmo(_f32(gm_latent), 5, mu=inv))
    return ops.hdr_tail(sdr_dec, gm_dec, 2, B, H, W, qmax=qmax, eps=eps, clamp=clamp, want=want)


def recompose(sdr_dec, gm_dec, qmax=99.0, eps=1 / 64, clamp=False, **kw):
    """Same tail for already-decoded NCHW images in [-1,1] (e.g. ``vae.decode(...)[0]``)."""
    B, _, H, W = sdr_dec.shape
    return ops.hdr_tail(sdr_dec.contiguous(), gm_dec.contiguous(), 0, B, H, W, qmax=qmax, eps=eps, clamp=clamp, **kw)


def to_host(out):
    return {k: v.cpu().numpy() for k, v in out.items()}


def _f32(x):
    x = x.contiguous()
    return x if x.dtype == torch.float32 else ops.cast(x, torch.float32)
