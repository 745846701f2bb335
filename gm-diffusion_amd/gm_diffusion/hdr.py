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
    # both latents go through the (shared) decoder as ONE batch of 2B: half the launches, fuller grids
    both = torch.cat([_f32(sdr_latent), _f32(gm_latent)], 0)
    dec, H, W = vae.decode_nhwc(ops.tmo(both, 5, mu=inv))  # [2B, H*W, 4] float32
    if vae.dtype == torch.float32:
        ops.check_split_range("decode_to_hdr: decoded images", dec, module=vae)  # the VAE's activations are the widest of the path
    return ops.hdr_tail(dec[:B], dec[B:], 2, B, H, W, qmax=qmax, eps=eps, clamp=clamp, want=want)


def recompose(sdr_dec, gm_dec, qmax=99.0, eps=1 / 64, clamp=False, **kw):
    """Same tail for already-decoded NCHW images in [-1,1] (e.g. ``vae.decode(...)[0]``)."""
    B, _, H, W = sdr_dec.shape
    return ops.hdr_tail(sdr_dec.contiguous(), gm_dec.contiguous(), 0, B, H, W, qmax=qmax, eps=eps, clamp=clamp, **kw)


def to_host(out):
    return {k: v.cpu().numpy() for k, v in out.items()}


def _f32(x):
    x = x.contiguous()
    return x if x.dtype == torch.float32 else ops.cast(x, torch.float32)


# ------------------------------------------------------------------------------------------------
# file writers: the reference uses cv2.imwrite("*.hdr") / PIL; cv2 is absent here
# ------------------------------------------------------------------------------------------------
def rgbe_scanlines(px, compression="rle"):
    """Bytes that follow the header of a Radiance picture for RGBE pixels ``px`` ([H, W, 4] uint8 host array): run-length
    framed scanlines ("rle": what OpenCV's encoder behind cv2.imwrite writes by default; host function gmd_rgbe_rle_encode of
    the C ABI) or flat pixels ("none": cv2's IMWRITE_HDR_COMPRESSION_NONE)."""
    import ctypes

    import numpy as np

    from ._native import check, lib
    px = np.ascontiguousarray(px, dtype=np.uint8)
    if px.ndim != 3 or px.shape[-1] != 4:
        raise ValueError("rgbe_scanlines expects [H, W, 4] bytes")
    if compression == "none":
        return px.tobytes()
    if compression != "rle":
        raise ValueError("compression must be 'rle' or 'none'")
    h, w = int(px.shape[0]), int(px.shape[1])
    cap = int(lib().gmd_rgbe_rle_bound(h, w))
    out = np.empty(max(cap, 1), np.uint8)
    n = ctypes.c_int64(0)
    check(lib().gmd_rgbe_rle_encode(px.ctypes.data, h, w, out.ctypes.data, cap, ctypes.addressof(n)), "gmd_rgbe_rle_encode")
    return out[: n.value].tobytes()


def save_hdr_image(hdr_file_rgb, path, compression="rle"):
    """Write one Radiance RGBE picture.  ``hdr_file_rgb``: [H,W,3] float32 device tensor, already divided by
    (qmax+1) and in RGB order (= ``out['hdr_file'][i]``); this is what the reference's ``save_hdr_image``
    (generate_hdr.py:27-30) hands to ``cv2.imwrite`` after its BGR swap.  Pixels are encoded on the device
    (gmd_rgbe_encode); scanlines are run-length framed on the host like OpenCV's writer does by default
    (``compression="none"`` writes them flat).  cv2 is absent in this image: header text and framing follow the published
    Radiance format, not a byte comparison with cv2's output.
    Negative values (possible with the unclamped Eq. 1) are stored as 0: RGBE has no sign."""
    x = hdr_file_rgb.contiguous()
    if x.dim() != 3 or x.shape[-1] != 3:
        raise ValueError("save_hdr_image expects [H, W, 3]")
    h, w = x.shape[0], x.shape[1]
    px = ops.rgbe_encode(_f32(x)).cpu().numpy()
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n")
        f.write(f"-Y {h} +X {w}\n".encode())
        f.write(rgbe_scanlines(px.reshape(h, w, 4), compression))


def save_png_u8(u8_rgb, path):
    """[H,W,3] uint8 device/host tensor -> PNG (generate_hdr.py:244-245 uses PIL the same way)."""
    from PIL import Image

    Image.fromarray(u8_rgb.cpu().numpy() if torch.is_tensor(u8_rgb) else u8_rgb).save(path)
