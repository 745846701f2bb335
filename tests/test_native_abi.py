"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/gmd_hip.h declares;
argument validation returns error codes without touching a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    from gm_diffusion import _native

    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    return _native


def test_header_and_binding_agree(native):
    hdr = open(os.path.join(ROOT, "include", "gmd_hip.h")).read()
    declared = set(re.findall(r"\b(gmd_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(native.SIGNATURES), declared ^ set(native.SIGNATURES)
    lib = native.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported"
    assert lib.gmd_abi_version() == native.ABI_VERSION
    m = re.search(r"#define GMD_ABI_VERSION (\d+)", hdr)
    assert int(m.group(1)) == native.ABI_VERSION


def test_argument_validation_without_gpu(native):
    lib = native.lib()
    assert lib.gmd_tmo(None, None, 4, 99, 1.0, 1.0, None) == 1  # GMD_ERR_INVALID
    assert b"kind" in lib.gmd_last_error()
    assert lib.gmd_tmo(None, None, 0, 0, 1.0, 1.0, None) == 0  # n == 0 is a no-op
    assert lib.gmd_quantize_u8(None, None, -1, None) == 1
    assert lib.gmd_latent_step(None, None, None, None, None, None, 1, 16, 0, 1.0, None, 0.0, 7, 1.0, 1.0, 1.0, 1.0, 0.0, None, None, None, None) == 1
    assert b"mode" in lib.gmd_last_error()
    assert lib.gmd_attention(None, None, None, None, native.GMD_F32, 1, 1, 40, 8, 8, 40, 40, 8, 40, 0, 0, 0, 0, 1.0, 0, None) == 3  # UNSUPPORTED
    assert lib.gmd_groupnorm_nsplit(4096) >= 1


def test_no_cpu_fallback():
    import torch

    from gm_diffusion import apply_gm_to_sdr, hip_ops
    from gm_diffusion._native import HipExtensionError

    with pytest.raises(HipExtensionError):
        apply_gm_to_sdr(torch.zeros(1, 3, 2, 2), torch.zeros(1, 3, 2, 2))
    with pytest.raises(HipExtensionError):
        hip_ops.gemm_nt(torch.zeros(64, 64), torch.zeros(64, 64))
    z = torch.zeros(1, 4, 8, 8)
    host_calls = [
        lambda: hip_ops.conv3x3(torch.zeros(1, 64, 64), torch.zeros(64, 576), 1, 8, 8),
        lambda: hip_ops.attention(torch.zeros(1, 64, 64), torch.zeros(1, 64, 64), torch.zeros(1, 64, 64), 2, 64, 1.0),
        lambda: hip_ops.groupnorm(torch.zeros(1, 64, 64), 1, 8, torch.ones(64), torch.zeros(64), 1e-5),
        lambda: hip_ops.layernorm(torch.zeros(4, 64), torch.ones(64), torch.zeros(64)),
        lambda: hip_ops.embedding_lookup(torch.zeros(1, 4, dtype=torch.long), torch.zeros(10, 8), torch.zeros(4, 8)),
        lambda: hip_ops.latent_step(z, z, 0, (1.0, 0.5, 1.0, 1.0, 0.0), False, 1.0),
        lambda: hip_ops.dpm_step(z, z, 1, (0.5, 0.9, 0.9, -0.1, -0.05, 0.0, 1.0, 0.0), False, 1.0),
        lambda: hip_ops.discretize_u16(torch.zeros(8)),
        lambda: hip_ops.rgbe_encode(torch.zeros(4, 3)),
    ]
    for call in host_calls:  # every front-end op refuses host tensors: there is no CPU path to fall back to
        with pytest.raises(HipExtensionError):
            call()
    from gm_diffusion.components import CLIPTextModel, UNet2DConditionModel

    with pytest.raises(HipExtensionError):
        UNet2DConditionModel(block_out_channels=(32, 32, 32, 32), cross_attention_dim=32, attention_head_dim=2, norm_num_groups=8).init_random(0)(
            torch.zeros(1, 4, 8, 8), 1, encoder_hidden_states=torch.zeros(1, 77, 32))
    with pytest.raises(HipExtensionError):
        CLIPTextModel(vocab_size=100, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2).init_random(0)(
            torch.zeros(1, 77, dtype=torch.long))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from gm_diffusion import _native

    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_native.HipExtensionError):
        _native.lib()


def test_export_surface_matches_reference():
    import gm_diffusion
    import gm_diffusion.pipelines as P
    import gm_diffusion.stage1 as S

    assert P.__all__ == ["StableDiffusionGMPipeline", "StableDiffusionDualUNetPipeline",
                         "StableDiffusionDualUNetImprovedPipeline", "rescale_noise_cfg", "retrieve_timesteps"]
    assert gm_diffusion.__all__ == ["RandomExposureAdjust", "apply_gm_to_sdr", "gamut_compress", "hard_clip_tmo",
                                    "linear_scale_tmo", "random_tmo_cuda", "tmo_cuda"]
    for n in ("RandomExposureAdjust", "apply_gm_to_sdr", "fix_mulog_tmo", "gamut_compress", "hard_clip_tmo",
              "linear_scale_tmo", "random_tmo_cuda", "tmo_cuda"):
        assert hasattr(S, n)
    assert issubclass(P.StableDiffusionDualUNetImprovedPipeline, P.StableDiffusionDualUNetPipeline)


def test_host_rgbe_run_length_encoder_equals_oracle():
    """gmd_rgbe_rle_encode is a HOST function of the C ABI (no device work): byte-identical to the oracle's restatement of the
    Radiance scanline framing on random, run-heavy, ragged and flat-width inputs; capacity is checked."""
    import ctypes

    import numpy as np

    from gm_diffusion import hdr
    from gm_diffusion._native import lib
    from oracle import hdr_ops as H
    rng = np.random.default_rng(11)
    for h, w in ((1, 8), (3, 53), (2, 128), (2, 129), (4, 300), (2, 7), (0, 16), (1, 40000)):
        px = rng.integers(0, 4, (h, w, 4), dtype=np.uint8) * rng.integers(0, 2, (h, w, 1), dtype=np.uint8)  # many runs
        if h and w > 200:
            px[0, 10:160, 2] = np.arange(150, dtype=np.uint8)  # more than 128 literals
        got = hdr.rgbe_scanlines(px)
        assert got == H.rgbe_rle_scanlines(px), (h, w)
        if h:
            assert np.array_equal(H.rgbe_rle_decode(got, h, w), px)
    assert hdr.rgbe_scanlines(px, "none") == px.tobytes()
    # wide, short, incompressible: the encoder's one-line scratch sits in the last W bytes of the bounded buffer and must stay
    # clear of the output (round 3's bound left 4*H + a few bytes of slack: for 1 x 1000 the last planes' literals overlapped it)
    for h, w in ((1, 1000), (2, 4096), (1, 32767)):
        px = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        px[:, 1:, :] += (px[:, 1:, :] == px[:, :-1, :]).astype(np.uint8)  # no two equal neighbours: literals only, the longest output
        got = hdr.rgbe_scanlines(px)
        assert len(got) == h * (4 + 4 * (w + (w + 127) // 128)) and len(got) + w <= lib().gmd_rgbe_rle_bound(h, w)
        assert got == H.rgbe_rle_scanlines(px), (h, w)
        assert np.array_equal(H.rgbe_rle_decode(got, h, w), px)
    px = np.zeros((2, 16, 4), np.uint8)
    out = np.zeros(8, np.uint8)
    n = ctypes.c_int64(0)
    assert lib().gmd_rgbe_rle_encode(px.ctypes.data, 2, 16, out.ctypes.data, 8, ctypes.addressof(n)) == 1  # GMD_ERR_INVALID
    assert b"smaller than gmd_rgbe_rle_bound" in lib().gmd_last_error()


def test_no_swizzled_packed_f32_next_to_an_exec_write():
    """Round-4 fault guard, wired into the suite: every kernel is compiled to gfx950 assembly and scanned for a
    swizzled packed-float32 instruction (`v_pk_{add,mul,fma}_f32 ... op_sel:[..]`) within six instructions of an EXEC write -- the
    pattern whose low result lost an addend in lanes 48..63 in 1 launch of 200-1500.  `make lint` keeps a
    stamp (build/lint.ok) that is redone whenever a source, a header or the lint itself changes, so this is a no-op after
    `__graft_entry__.build()` and ~2.5 minutes of hipcc on a tree whose kernels changed since."""
    import subprocess

    csrc = os.path.join(ROOT, "gm-diffusion_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "lint"], capture_output=True, text=True)
    assert r.returncode == 0, "assembly lint failed:\n" + r.stdout[-3000:] + r.stderr[-2000:]
    assert os.path.exists(os.path.join(csrc, "build", "lint.ok"))


def test_lint_recognises_the_faulty_pattern(tmp_path):
    """The lint must actually fire on the instruction sequence of the faulty round-4 build (its regular expressions, not hipcc)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("lint_pk", os.path.join(ROOT, "tools", "lint_pk_opsel_exec.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    bad = ["v_pk_add_f32 v[82:83], v[118:119], v[84:85] op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_add_f32 v[84:85], v[116:117], v[84:85]",
           "v_cmp_gt_u32_e32 vcc, s0, v123", "s_and_saveexec_b64 s[0:1], vcc"]
    assert m.PK.match(bad[0]) and not m.PK.match(bad[1]) and m.EXEC_W.match(bad[3]) and not m.EXEC_W.match(bad[2])
    assert not m.PK.match("v_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel_hi:[1,0]")  # the low-register broadcast is not the pattern
