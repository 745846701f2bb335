"""
Oracle: numpy restatement of the reference's HDR / gain-map elementwise ops.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Each function cites the reference lines it follows (paths relative to
/root/reference).  All float math is float32, as in the reference (torch
float32 tensors / numpy float32 arrays from ``.float().numpy()``).

Pinned by tests/golden/hdr_ops_*.npz, produced by oracle/make_golden.py from
the reference's own tone_mapping.py / augmentations.py loaded by file path.
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32


def _f32(x):
    return np.asarray(x, dtype=F32)


# --------------------------------------------------------------------------
# gm_diffusion/stage1/tone_mapping.py
# --------------------------------------------------------------------------
def linear_scale_tmo(img, qmax):
    """tone_mapping.py:14-18  ``img / (qmax + 1)``"""
    return _f32(img) / F32(qmax + 1)


def hard_clip_tmo(img, qmax=None):
    """tone_mapping.py:21-26  ``clamp(img, 0, 1)`` (qmax ignored)."""
    return np.clip(_f32(img), F32(0), F32(1))


def fix_mulog_tmo(img, qmax):
    """tone_mapping.py:29-36  ``clamp(log1p(500 * img/(qmax+1)) / log1p(500), 0, 1)``"""
    x = _f32(img) / F32(qmax + 1)
    tm = np.log1p(F32(500) * x) / F32(math.log1p(500))
    return np.clip(tm, F32(0), F32(1))


def tmo_cuda(img):
    """tone_mapping.py:39-47  ``x = clamp(img/10, 0, 1); log1p(5000 x) / log1p(5000)``"""
    x = np.clip(_f32(img) / F32(10), F32(0), F32(1))
    if not np.all((0 <= x) & (x <= 1)):
        raise ValueError("HDR image values should be in the range [0, 1]")
    return np.log1p(F32(5000.0) * x) / F32(math.log1p(5000.0))


def mulog_tmo(img, qmax, mu):
    """tone_mapping.py:50-57 ``random_tmo_cuda`` with the random ``mu`` made explicit."""
    x = _f32(img) / F32(qmax + 1)
    tm = np.log1p(F32(mu) * x) / F32(math.log1p(mu))
    return np.clip(tm, F32(0), F32(1))


def apply_gm_to_sdr(gm, sdr, qmax=9, eps=1 / 64, clamp=True):
    """
    Eq. 1 gain-map recomposition.

    clamp=True : torch variant, tone_mapping.py:60-71 (output clamped to [0, qmax+1]).
    clamp=False: numpy variant copied into every experiment script,
                 scripts/inference/experiments/formal_improved.py:34-45 (no clamp).
    """
    sdr_linear = np.power(np.clip(_f32(sdr), F32(0), F32(1)), F32(2.2))
    hdr = (sdr_linear + F32(eps)) * (F32(1) + _f32(gm) * F32(qmax)) - F32(eps)
    if clamp:
        hdr = np.clip(hdr, F32(0), F32(qmax + 1))
    return hdr.astype(F32)


# BT.2020 -> BT.709, row-major constants of tone_mapping.py:78-84
GAMUT_2020_TO_709 = np.array(
    [
        [1.660491, -0.587641, -0.072850],
        [-0.124550, 1.132900, -0.008349],
        [-0.018151, -0.100579, 1.118730],
    ],
    dtype=F32,
)


def gamut_compress(img_nchw):
    """tone_mapping.py:74-90  per-pixel ``pixel @ M.T`` on NCHW input, then clamp(0,1).

    out[c] = sum_k in[k] * M[c][k]  (``matmul(img_nhwc, M.t())``)."""
    x = _f32(img_nchw)
    nhwc = np.transpose(x, (0, 2, 3, 1))
    out = np.matmul(nhwc, GAMUT_2020_TO_709.T.copy())
    out = np.transpose(out, (0, 3, 1, 2))
    return np.clip(out, F32(0), F32(1)).astype(F32)


# --------------------------------------------------------------------------
# quantisers
# --------------------------------------------------------------------------
def discretize_to_uint16(img):
    """augmentations.py:38-41  ``clamp(img*65535, 0, 65535).round() / 65535``
    (torch.round = round-half-to-even); returned as float32 like the reference."""
    max_int = F32(2**16 - 1)
    q = np.rint(np.clip(_f32(img) * max_int, F32(0), max_int))  # rint = half-to-even
    return (q / max_int).astype(F32)


def quantize_u16_codes(img):
    """Integer codes of :func:`discretize_to_uint16` (what the float value encodes)."""
    max_int = F32(2**16 - 1)
    return np.rint(np.clip(_f32(img) * max_int, F32(0), max_int)).astype(np.uint16)


def quantize_u8_trunc(img01):
    """scripts/inference/generate_hdr.py:244-245  ``(x * 255).astype(np.uint8)`` --
    float32 multiply then C truncation toward zero (inputs are in [0,1])."""
    return (_f32(img01) * F32(255)).astype(np.uint8)


def denorm_clamp(x):
    """generate_hdr.py:227 / stable_diffusion_gm.py:606  ``(x / 2 + 0.5).clamp(0, 1)``"""
    return np.clip(_f32(x) / F32(2) + F32(0.5), F32(0), F32(1))


def save_hdr_scale(hdr, qmax):
    """generate_hdr.py:27-30: ``hdr / (qmax+1)`` -> float32 -> channel order [2,1,0] (BGR)."""
    return (_f32(hdr) / F32(qmax + 1)).astype(F32)[..., [2, 1, 0]]


# --------------------------------------------------------------------------
# the Stage-3 tail as the CLI composes it (generate_hdr.py:225-265,
# formal_improved.py:272-303): decode outputs -> images -> Eq.1 -> /(qmax+1)
# --------------------------------------------------------------------------
def hdr_tail(sdr_dec_nchw, gm_dec_nchw, qmax=99, eps=1 / 64, clamp=False):
    """
    sdr_dec/gm_dec: VAE decoder outputs (B,3,H,W) in [-1,1].
    Returns dict of NHWC arrays: sdr, gm (float32 in [0,1]), sdr_u8, gm_u8
    (truncated PNG bytes), hdr (Eq.1, float32), hdr_file (= hdr/(qmax+1), RGB order).
    """
    sdr = np.transpose(denorm_clamp(sdr_dec_nchw), (0, 2, 3, 1))
    gm = np.transpose(denorm_clamp(gm_dec_nchw), (0, 2, 3, 1))
    hdr = apply_gm_to_sdr(gm, sdr, qmax=qmax, eps=eps, clamp=clamp)
    return {
        "sdr": sdr,
        "gm": gm,
        "sdr_u8": quantize_u8_trunc(sdr),
        "gm_u8": quantize_u8_trunc(gm),
        "hdr": hdr,
        "hdr_file": (hdr / F32(qmax + 1)).astype(F32),
    }


def stage1_chain(gm_nchw, sdr_nchw, qmax=49):
    """scripts/stage1/train_vqgan_lora.py:1133-1141: Eq.1 (torch, clamped) ->
    fix_mulog_tmo -> gamut_compress."""
    hdr = apply_gm_to_sdr(gm_nchw, sdr_nchw, qmax=qmax, clamp=True)
    return gamut_compress(fix_mulog_tmo(hdr, qmax))


def rgbe_encode(rgb):
    """Radiance RGBE pixels (G. Ward's ``float2rgbe``, the encoder inside OpenCV's HDR writer that the reference
    calls through ``cv2.imwrite`` at generate_hdr.py:27-30; cv2 is not installed here, so this restates the published
    algorithm): v = max(r,g,b); v < 1e-32 -> (0,0,0,0); else (m, e) = frexp(v), s = m*256/v,
    bytes = trunc(c*s), exponent byte = e + 128.  Negative components are stored as 0 (RGBE has no sign)."""
    x = np.maximum(_f32(rgb), F32(0))
    v = x.max(axis=-1)
    m, e = np.frexp(v)
    ok = v >= F32(1e-32)
    s = (m.astype(F32) * F32(256.0) / np.where(ok, v, F32(1))).astype(F32)
    out = np.zeros(x.shape[:-1] + (4,), np.uint8)
    out[..., :3] = np.where(ok[..., None], (x * s[..., None]).astype(np.int32), 0).astype(np.uint8)
    out[..., 3] = np.where(ok, e + 128, 0).astype(np.uint8)
    return out


def rgbe_rle_scanlines(px):
    """Run-length framing of RGBE scanlines as the Radiance format defines it and rgbe.c's RGBE_WritePixels_RLE /
    RGBE_WriteBytes_RLE write it -- the writer inside OpenCV's HDR encoder, which the reference reaches through cv2.imwrite at
    generate_hdr.py:27-30 (cv2 absent here: restated from the published format, unpinned against cv2's bytes).  ``px``:
    [H, W, 4] uint8.  Pure-Python loops: small cases only."""
    px = np.ascontiguousarray(px, np.uint8)
    h, w = px.shape[0], px.shape[1]
    if w < 8 or w > 0x7FFF:
        return px.tobytes()
    out = bytearray()
    for y in range(h):
        out += bytes((2, 2, w >> 8, w & 0xFF))
        for c in range(4):
            data = px[y, :, c].tolist()
            cur = 0
            while cur < w:
                beg, run, old = cur, 0, 0
                while run < 4 and beg < w:
                    beg += run
                    old = run
                    run = 1
                    while beg + run < w and run < 127 and data[beg] == data[beg + run]:
                        run += 1
                if old > 1 and old == beg - cur:
                    out += bytes((128 + old, data[cur]))
                    cur = beg
                while cur < beg:
                    m = min(beg - cur, 128)
                    out.append(m)
                    out += bytes(data[cur:cur + m])
                    cur += m
                if run >= 4:
                    out += bytes((128 + run, data[beg]))
                    cur += run
    return bytes(out)


def rgbe_rle_decode(buf, h, w):
    """Reader for :func:`rgbe_rle_scanlines` (the adaptive run-length scheme any Radiance reader implements): -> [H, W, 4] uint8."""
    buf = bytes(buf)
    if w < 8 or w > 0x7FFF:
        return np.frombuffer(buf, np.uint8).reshape(h, w, 4).copy()
    out = np.zeros((h, w, 4), np.uint8)
    p = 0
    for y in range(h):
        assert buf[p] == 2 and buf[p + 1] == 2 and (buf[p + 2] << 8 | buf[p + 3]) == w, "bad scanline header"
        p += 4
        for c in range(4):
            x = 0
            while x < w:
                n = buf[p]
                p += 1
                if n > 128:
                    n -= 128
                    out[y, x:x + n, c] = buf[p]
                    p += 1
                else:
                    assert n > 0
                    out[y, x:x + n, c] = np.frombuffer(buf[p:p + n], np.uint8)
                    p += n
                x += n
            assert x == w, "run crosses the end of a scanline"
    assert p == len(buf)
    return out


def rgbe_decode(px):
    """Inverse of :func:`rgbe_encode` (Ward's ``rgbe2float`` without the +0.5 bias OpenCV also omits)."""
    px = np.asarray(px)
    f = np.ldexp(F32(1.0), px[..., 3].astype(np.int32) - (128 + 8)).astype(F32)
    return np.where(px[..., 3:4] == 0, F32(0), px[..., :3].astype(F32) * f[..., None]).astype(F32)
