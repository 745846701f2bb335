"""CPU: the numpy oracle of the HDR ops against golden vectors produced by the REFERENCE's own
tone_mapping.py / augmentations.py (tests/golden/hdr_ops_reference.npz, oracle/make_golden.py)."""
import os

import numpy as np
import pytest

from oracle import hdr_ops as H


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "hdr_ops_reference.npz"))


def _close(a, b, ulps=2, floor=2.4e-7):
    """|a-b| <= ulps * ulp(max(|a|,|b|)) + floor.  The floor (2 ulp of 1.0) covers sums with cancellation
    (gamut matrix, log ratio) whose ABSOLUTE error is one rounding of an O(1) intermediate."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    assert a.shape == b.shape
    tol = ulps * np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32)).astype(np.float64) + floor
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    assert np.all(d <= tol), float(d.max())


def test_denorm_and_linear_ops_exact(g):
    assert np.array_equal(H.denorm_clamp(g["sdr_dec"]), g["sdr"])
    assert np.array_equal(H.denorm_clamp(g["gm_dec"]), g["gm"])
    for q in (9, 49, 99):
        hdr = g[f"apply_gm_to_sdr_q{q}"]
        assert np.array_equal(H.linear_scale_tmo(hdr, q), g[f"linear_scale_tmo_q{q}"])
        assert np.array_equal(H.hard_clip_tmo(hdr, q), g[f"hard_clip_tmo_q{q}"])


@pytest.mark.parametrize("q", [9, 49, 99])
def test_eq1_and_tmos_within_ulps(g, q):
    # float32 pow/log1p differ by <= 1-2 ulp between numpy and the torch the reference ran on
    _close(H.apply_gm_to_sdr(g["gm"], g["sdr"], qmax=q), g[f"apply_gm_to_sdr_q{q}"], ulps=4)
    _close(H.fix_mulog_tmo(g[f"apply_gm_to_sdr_q{q}"], q), g[f"fix_mulog_tmo_q{q}"])
    _close(H.gamut_compress(g[f"fix_mulog_tmo_q{q}"]), g[f"stage1_chain_q{q}"], ulps=4)


def test_defaults_and_other_tmos(g):
    _close(H.apply_gm_to_sdr(g["gm"], g["sdr"]), g["apply_gm_to_sdr_default"], ulps=4)
    _close(H.tmo_cuda(g["apply_gm_to_sdr_q9"]), g["tmo_cuda"])
    _close(H.gamut_compress(g["sdr"]), g["gamut_compress"], ulps=4)
    _close(H.gamut_compress(g["apply_gm_to_sdr_q9"]), g["gamut_compress_hdr"], ulps=4)
    _close(H.mulog_tmo(g["apply_gm_to_sdr_q49"], 49, float(g["random_tmo_mu"])), g["random_tmo_cuda_q49"])
    _close(H.apply_gm_to_sdr(np.clip(g["edge"][::-1], 0, 1), g["edge"], qmax=99), g["edge_apply_q99"], ulps=4)


def test_uint16_discretiser_bit_exact(g):
    assert np.array_equal(H.discretize_to_uint16(g["u16_in"]), g["discretize_to_uint16"])
    codes = H.quantize_u16_codes(g["u16_in"])
    assert codes.dtype == np.uint16
    assert np.array_equal((codes.astype(np.float32) / np.float32(65535)), g["discretize_to_uint16"])
    # round-half-to-even on exact .5 codes
    x = (np.array([0.5, 1.5, 2.5, 3.5], np.float32) / np.float32(65535)).astype(np.float32)
    got = H.quantize_u16_codes(x)
    assert set(got.tolist()) <= {0, 1, 2, 3, 4}


def test_uint8_truncation_and_variants():
    x = np.array([0.0, 0.999, 1.0, 254.9 / 255, 0.5], np.float32)
    assert H.quantize_u8_trunc(x).tolist() == [0, 254, 255, 254, 127]
    sdr = np.random.default_rng(0).random((1, 4, 5, 3), dtype=np.float32)
    gm = np.random.default_rng(1).random((1, 4, 5, 3), dtype=np.float32)
    clamped = H.apply_gm_to_sdr(gm, sdr, qmax=9, clamp=True)
    raw = H.apply_gm_to_sdr(gm, sdr, qmax=9, clamp=False)
    assert np.array_equal(clamped, np.clip(raw, 0, 10))
    assert raw.min() >= -1 / 64


def test_tail_composition_shapes():
    rng = np.random.default_rng(2)
    a = (rng.random((2, 3, 6, 8), dtype=np.float32) * 2.4 - 1.2).astype(np.float32)
    b = (rng.random((2, 3, 6, 8), dtype=np.float32) * 2.4 - 1.2).astype(np.float32)
    t = H.hdr_tail(a, b, qmax=99)
    assert t["hdr"].shape == (2, 6, 8, 3) and t["sdr_u8"].dtype == np.uint8
    assert np.array_equal(t["hdr_file"], (t["hdr"] / np.float32(100)).astype(np.float32))
    assert np.array_equal(H.save_hdr_scale(t["hdr"], 99), t["hdr_file"][..., [2, 1, 0]])


def test_rgbe_known_answers_and_roundtrip():
    x = np.array([[1, 1, 1], [0.5, 0.25, 0.125], [0, 0, 0], [1e-33, 0, 0], [3.7, 0.2, -0.01], [255.9, 1, 1]], np.float32)
    e = H.rgbe_encode(x)
    assert e.tolist() == [[128, 128, 128, 129], [128, 64, 32, 128], [0, 0, 0, 0], [0, 0, 0, 0], [236, 12, 0, 130], [255, 1, 1, 136]]
    rng = np.random.default_rng(0)
    y = (rng.random((64, 64, 3), dtype=np.float32) * 40).astype(np.float32)
    d = H.rgbe_decode(H.rgbe_encode(y))
    # 8-bit mantissa relative to the pixel maximum, truncation: error < max/128 per component
    assert np.all(y - d >= 0) and np.all(y - d <= y.max(-1, keepdims=True) / 128 + 1e-6)


def test_rgbe_run_length_scanlines_known_answers_and_roundtrip():
    """The scanline framing of Radiance pictures (oracle restatement of rgbe.c's writer; cv2 absent): hand-checked byte strings
    for the run / literal / short-run cases, flat output outside 8 <= width <= 32767, and decode(encode(x)) == x."""
    w = 16
    line = np.zeros((1, w, 4), np.uint8)
    line[0, :, 0] = [7] * 16                                        # one long run
    line[0, :, 1] = list(range(16))                                 # literals only
    line[0, :, 2] = [1, 1, 5, 5, 5, 5, 5, 2, 3, 3, 3, 9, 9, 9, 9, 4]  # short run, long run, literals (a 3-run stays literal), run, tail
    line[0, :, 3] = [128] * 5 + [129] * 11
    got = H.rgbe_rle_scanlines(line)
    want = bytes([2, 2, 0, 16]) + bytes([128 + 16, 7]) + bytes([16] + list(range(16))) \
        + bytes([128 + 2, 1, 128 + 5, 5, 4, 2, 3, 3, 3, 128 + 4, 9, 1, 4]) + bytes([128 + 5, 128, 128 + 11, 129])
    assert got == want
    assert np.array_equal(H.rgbe_rle_decode(got, 1, w), line)
    rng = np.random.default_rng(5)
    for h, w in ((3, 8), (2, 200), (5, 131), (1, 300)):
        px = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        px[:, w // 3: w // 3 + 140, 3] = 130   # a run longer than 127 and literals longer than 128 in one line
        px[0, :, 0] = 9
        enc = H.rgbe_rle_scanlines(px)
        assert np.array_equal(H.rgbe_rle_decode(enc, h, w), px)
    small = rng.integers(0, 256, (4, 7, 4), dtype=np.uint8)
    assert H.rgbe_rle_scanlines(small) == small.tobytes()  # width < 8: flat
