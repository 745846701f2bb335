"""GPU parity tests of the individual HIP kernels (through the C ABI via gm_diffusion.hip_ops)
against plain PyTorch fp32/fp64 references and the oracle's golden vectors."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def ops():
    from gm_diffusion import hip_ops

    return hip_ops


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_err(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max())


HALF = (torch.bfloat16, torch.float16)


def tol(dtype):
    return 2e-5 if dtype == torch.float32 else (1.5e-2 if dtype == torch.bfloat16 else 2e-3)


# ---------------------------------------------------------------------------------------------
# HDR tail
# ---------------------------------------------------------------------------------------------
def test_hdr_ops_against_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "hdr_ops_reference.npz"))
    o = ops()
    sdr, gm = torch.from_numpy(g["sdr"]).to(DEV), torch.from_numpy(g["gm"]).to(DEV)
    for q in (9, 49, 99):
        ref = torch.from_numpy(g[f"apply_gm_to_sdr_q{q}"])
        got = o.apply_gm_to_sdr(gm, sdr, qmax=q)
        assert rel_err(got, ref) < 2e-7 and max_err(got, ref) <= 4e-6 * (q + 1)
        hdr = ref.to(DEV)
        assert max_err(o.tmo(hdr, 2, qmax=q, mu=500.0), torch.from_numpy(g[f"fix_mulog_tmo_q{q}"])) <= 2.4e-7
        assert torch.equal(o.tmo(hdr, 0, qmax=q).cpu(), torch.from_numpy(g[f"linear_scale_tmo_q{q}"]))
        assert torch.equal(o.tmo(hdr, 1).cpu(), torch.from_numpy(g[f"hard_clip_tmo_q{q}"]))
        assert max_err(o.stage1_chain(gm, sdr, q), torch.from_numpy(g[f"stage1_chain_q{q}"])) <= 1e-6
    assert max_err(o.tmo(torch.from_numpy(g["apply_gm_to_sdr_q9"]).to(DEV), 3), torch.from_numpy(g["tmo_cuda"])) <= 2.4e-7
    assert max_err(o.gamut_compress(sdr), torch.from_numpy(g["gamut_compress"])) <= 2.4e-7
    mu = float(g["random_tmo_mu"])
    assert max_err(o.tmo(torch.from_numpy(g["apply_gm_to_sdr_q49"]).to(DEV), 2, qmax=49, mu=mu),
                   torch.from_numpy(g["random_tmo_cuda_q49"])) <= 2.4e-7
    # integer / exactly-rounded ops: bit exact
    u16 = o.discretize_u16(torch.from_numpy(g["u16_in"]).to(DEV))
    assert torch.equal(u16.cpu(), torch.from_numpy(g["discretize_to_uint16"]))
    # denorm: exact
    assert torch.equal(o.tmo(torch.from_numpy(g["sdr_dec"]).to(DEV), 4).cpu(), torch.from_numpy(g["sdr"]))


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_hdr_tail_fused_vs_oracle(layout, dtype):
    from oracle import hdr_ops as H

    o = ops()
    g = torch.Generator().manual_seed(3)
    B, Hh, W = 2, 20, 36
    sdr_dec = (torch.rand(B, 3, Hh, W, generator=g) * 2.4 - 1.2).to(dtype)
    gm_dec = (torch.rand(B, 3, Hh, W, generator=g) * 2.4 - 1.2).to(dtype)
    ref = H.hdr_tail(sdr_dec.float().numpy(), gm_dec.float().numpy(), qmax=99, clamp=False)

    def lay(x):
        if layout == 0:
            return x.contiguous()
        n = x.permute(0, 2, 3, 1)
        if layout == 2:
            n = torch.cat([n, torch.full_like(n[..., :1], 7.0)], -1)
        return n.reshape(B, Hh * W, -1).contiguous()

    out = o.hdr_tail(lay(sdr_dec).to(DEV), lay(gm_dec).to(DEV), layout, B, Hh, W, qmax=99.0, clamp=False)
    assert torch.equal(out["sdr"].cpu(), torch.from_numpy(ref["sdr"]))
    assert torch.equal(out["gm"].cpu(), torch.from_numpy(ref["gm"]))
    assert torch.equal(out["sdr_u8"].cpu(), torch.from_numpy(ref["sdr_u8"]))  # integer: bit exact
    assert torch.equal(out["gm_u8"].cpu(), torch.from_numpy(ref["gm_u8"]))
    assert rel_err(out["hdr"], torch.from_numpy(ref["hdr"])) < 2e-7
    assert rel_err(out["hdr_file"], torch.from_numpy(ref["hdr_file"])) < 2e-7
    # the u16 quantiser is exact given ITS float input: check against the kernel's own hdr_file
    codes = H.quantize_u16_codes(out["hdr_file"].cpu().numpy())
    assert np.array_equal(out["hdr_u16"].cpu().numpy(), codes)
    # and report (not gate) the end-to-end mismatch caused by <=1ulp powf differences
    mism = (out["hdr_u16"].cpu().numpy() != H.quantize_u16_codes(ref["hdr_file"])).mean()
    assert mism < 0.02
    # clamped variant (torch apply_gm_to_sdr)
    out_c = o.hdr_tail(lay(sdr_dec).to(DEV), lay(gm_dec).to(DEV), layout, B, Hh, W, qmax=9.0, clamp=True, want=("hdr",))
    ref_c = H.apply_gm_to_sdr(ref["gm"], ref["sdr"], qmax=9, clamp=True)
    assert rel_err(out_c["hdr"], torch.from_numpy(ref_c)) < 2e-7


def test_hdr_tail_vec4_path_bit_identical_to_generic():
    """The four-pixels-per-thread fast path (float32 [B,HW,4] input, 16-byte accesses) against the generic kernel on the same
    pixels given as packed RGB (layout 1) and with a pixel count that is not a multiple of 4 (generic again)."""
    o = ops()
    g = torch.Generator().manual_seed(13)
    B, Hh, W = 3, 24, 40
    s3 = (torch.rand(B, Hh * W, 3, generator=g) * 2.6 - 1.3)
    g3 = (torch.rand(B, Hh * W, 3, generator=g) * 2.6 - 1.3)
    pad = lambda x: torch.cat([x, torch.full_like(x[..., :1], -3.0)], -1).contiguous()
    fast = o.hdr_tail(pad(s3).to(DEV), pad(g3).to(DEV), 2, B, Hh, W, qmax=99.0)
    slow = o.hdr_tail(s3.to(DEV), g3.to(DEV), 1, B, Hh, W, qmax=99.0)
    assert set(fast) == set(slow) and len(fast) == 7
    for k in fast:
        assert torch.equal(fast[k], slow[k]), k
    # odd pixel count -> generic kernel for layout 2 as well
    odd = o.hdr_tail(pad(s3)[:1, :957].contiguous().to(DEV), pad(g3)[:1, :957].contiguous().to(DEV), 2, 1, 1, 957, qmax=99.0, clamp=True)
    ref = o.hdr_tail(s3[:1, :957].contiguous().to(DEV), g3[:1, :957].contiguous().to(DEV), 1, 1, 1, 957, qmax=99.0, clamp=True)
    for k in odd:
        assert torch.equal(odd[k], ref[k]), k


def test_quantisers_bit_exact_large():
    o = ops()
    from oracle import hdr_ops as H

    x = torch.rand(1 << 20, generator=torch.Generator().manual_seed(5)) * 1.1 - 0.05
    assert np.array_equal(o.discretize_u16(x.to(DEV), codes=True)[1].cpu().numpy(), H.quantize_u16_codes(x.numpy()))
    x01 = x.clamp(0, 1)
    assert np.array_equal(o.quantize_u8(x01.to(DEV)).cpu().numpy(), H.quantize_u8_trunc(x01.numpy()))
    # idempotence of the discretiser
    d1 = o.discretize_u16(x.to(DEV))
    assert torch.equal(o.discretize_u16(d1), d1)


def test_hdr_empty_inputs():
    o = ops()
    e = torch.empty(0, dtype=torch.float32, device=DEV)
    assert o.tmo(e, 1).numel() == 0 and o.discretize_u16(e).numel() == 0 and o.quantize_u8(e).numel() == 0


# ---------------------------------------------------------------------------------------------
# latent step / pack / unpack
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("do_cfg", [False, True])
def test_latent_step_bit_exact_vs_torch_pndm(do_cfg):
    """Drive the product PNDM scheduler on device tensors and the oracle PNDM on CPU tensors with
    identical eps; every step must agree bit for bit."""
    from gm_diffusion.components import PNDMScheduler
    from oracle import schedulers as OS

    B, shape = 3, (3, 4, 8, 8)
    g = torch.Generator().manual_seed(11)
    sp = PNDMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", skip_prk_steps=True, steps_offset=1)
    so = OS.PNDMScheduler()
    sp.set_timesteps(10)
    so.set_timesteps(10)
    assert torch.equal(sp.timesteps, so.timesteps)
    x_cpu = torch.randn(shape, generator=g)
    x_dev = x_cpu.to(DEV)
    gs = 7.5
    for t in so.timesteps:
        raw = torch.randn((2 * B if do_cfg else B,) + shape[1:], generator=g)
        if do_cfg:
            u, c = raw.chunk(2)
            eps = u + gs * (c - u)
        else:
            eps = raw
        a = so.alphas_cumprod[t].view(-1, 1, 1, 1)
        x0_ref = (x_cpu - (1 - a).sqrt() * eps) / a.sqrt()
        x_cpu = so.step(eps, t, x_cpu, return_dict=False)[0]
        x_dev, x0 = sp.fused_step(raw.to(DEV), t, x_dev, do_cfg, gs, want_x0=True)
        assert torch.equal(x0.cpu(), x0_ref), f"x0 differs at t={int(t)}"
        assert torch.equal(x_dev.cpu(), x_cpu), f"x_prev differs at t={int(t)}"


def test_pndm_step_protocol_device_equals_host():
    from gm_diffusion.components import PNDMScheduler

    kw = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", skip_prk_steps=True, steps_offset=1)
    sd, sh = PNDMScheduler(**kw), PNDMScheduler(**kw)
    sd.set_timesteps(7)
    sh.set_timesteps(7)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 4, 8, 8, generator=g)
    xd = x.to(DEV)
    for t in sh.timesteps:
        e = torch.randn(2, 4, 8, 8, generator=g)
        x = sh.step(e, t, x, return_dict=False)[0]
        xd = sd.step(e.to(DEV), t, xd, return_dict=False)[0]
        assert torch.equal(xd.cpu(), x)


def test_cfg_rescale_matches_torch():
    from oracle import pipelines as OP

    o = ops()
    g = torch.Generator().manual_seed(4)
    raw = torch.randn(6, 4, 16, 16, generator=g)
    u, c = raw.chunk(2)
    cfg = u + 5.0 * (c - u)
    ref = OP.rescale_noise_cfg(cfg, c, 0.7)
    ratio = o.cfg_std_ratio(raw.to(DEV), 5.0)
    x = torch.randn(3, 4, 16, 16, generator=g)
    eps, _, _ = o.latent_step(raw.to(DEV), x.to(DEV), 0, (1.0, 0.1, 1.0, 1.0, 0.0), True, 5.0, ratio=ratio, guidance_rescale=0.7)
    assert rel_err(eps, ref) < 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_pack_unpack(dtype):
    o = ops()
    g = torch.Generator().manual_seed(1)
    a, b = torch.randn(2, 4, 6, 10, generator=g), torch.randn(2, 4, 6, 10, generator=g)
    cp = 64 if dtype in HALF else 16
    out = o.pack_unet_input(a.to(DEV), b.to(DEV), 2, cp, dtype).cpu().float()
    ref = torch.cat([a, b], 1).permute(0, 2, 3, 1).reshape(2, 60, 8).to(dtype).float()
    assert out.shape == (4, 60, cp)
    assert torch.equal(out[:2, :, :8], ref) and torch.equal(out[2:, :, :8], ref)
    assert float(out[:, :, 8:].abs().max()) == 0.0
    single = o.pack_unet_input(a.to(DEV), None, 1, cp, dtype)
    back = o.unpack_nchw(single, 2, 4, 6, 10).cpu()
    assert torch.equal(back, a.to(dtype).float())


# ---------------------------------------------------------------------------------------------
# normalisation / elementwise
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,HW,C,G", [(2, 64, 320, 32), (1, 4096, 320, 32), (2, 256, 1280, 32), (1, 64, 2560, 32),
                                      (2, 100, 128, 32), (1, 16, 960, 32), (3, 37, 64, 8), (1, 4096, 960, 32), (2, 1024, 1920, 32), (1, 65536, 128, 32)])
def test_groupnorm(dtype, B, HW, C, G):
    o = ops()
    g = torch.Generator().manual_seed(C + HW)
    x = (torch.randn(B, HW, C, generator=g) * 2 + 0.5).to(dtype)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    xn = x.float().permute(0, 2, 1).reshape(B, C, HW, 1)
    for silu in (False, True):
        ref = F.group_norm(xn, G, gamma, beta, 1e-5)
        ref = F.silu(ref) if silu else ref
        ref = ref.reshape(B, C, HW).permute(0, 2, 1)
        got = o.groupnorm(x.to(DEV), B, G, gamma.to(DEV), beta.to(DEV), 1e-5, silu=silu)  # fused single launch when it fits
        assert rel_err(got.float(), ref) < (2e-6 if dtype == torch.float32 else 6e-3)
        got2 = o.groupnorm_split(x.to(DEV), B, G, gamma.to(DEV), beta.to(DEV), 1e-5, silu=silu)  # partial + finalize + apply
        assert rel_err(got2.float(), ref) < (2e-6 if dtype == torch.float32 else 6e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,HW,C,expect_u", [(1, 4163, 320, 2), (4, 4100, 320, 4), (5, 4100, 320, 6), (8, 4100, 320, 10), (3, 8200, 960, 10), (2, 3300, 640, 4)])
def test_groupnorm_apply_pass_every_vector_count_and_ragged_ends(dtype, B, HW, C, expect_u):
    """Round 5: the apply pass of the split-statistics GroupNorm is loop-free -- a workgroup owns 256 * U consecutive 16-byte vectors
    of a sample (U in {2, 4, 6, 10} by tensor size: four kernel instantiations), requests them in two halves around the statistics
    fold and stores through a range-checked buffer descriptor.  Every U, with sample sizes that are NOT a multiple of the span (the
    last workgroup's clamped loads / dropped stores) and a row count that is not a multiple of anything, against float64 torch."""
    o = ops()
    V = 8 if dtype in HALF else 4
    u = (HW * (C // V) * B) // (512 * 256)
    U = 10 if u >= 10 else 6 if u >= 6 else 4 if u >= 4 else 2  # gn_apply_span_u (csrc/norm.hip)
    if dtype in HALF:  # the 16-bit cases walk all four instantiations, each with a ragged last workgroup (float32: twice the vectors)
        assert U == expect_u and (HW * (C // V)) % (256 * U) != 0
    g = torch.Generator().manual_seed(B * HW + C)
    x = (torch.randn(B, HW, C, generator=g) * 1.5 - 0.25).to(dtype)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    for silu in (True, False):
        ref = F.group_norm(x.double().permute(0, 2, 1).reshape(B, C, HW, 1), 32, gamma.double(), beta.double(), 1e-5)
        ref = (F.silu(ref) if silu else ref).reshape(B, C, HW).permute(0, 2, 1)
        got = o.groupnorm(x.to(DEV), B, 32, gamma.to(DEV), beta.to(DEV), 1e-5, silu=silu)
        assert got.shape == x.shape and rel_err(got.double().cpu(), ref) < (3e-6 if dtype == torch.float32 else 6e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,C", [(5, 320), (130, 640), (64, 1280), (3, 64), (9, 2048), (4099, 320), (1031, 640), (32768, 320)])
def test_layernorm(dtype, rows, C):
    o = ops()
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, C, generator=g) * 3 - 1).to(dtype)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5)
    got = o.layernorm(x.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-5)
    assert rel_err(got.float(), ref) < (2e-6 if dtype == torch.float32 else 6e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_geglu_concat_cast_temb_softmax(dtype):
    o = ops()
    g = torch.Generator().manual_seed(8)
    x = torch.randn(37, 2 * 1280, generator=g).to(dtype)
    h, gate = x.float().chunk(2, -1)
    assert rel_err(o.geglu(x.to(DEV)).float(), h * F.gelu(gate)) < tol(dtype)
    a, b = torch.randn(33, 640, generator=g).to(dtype), torch.randn(33, 320, generator=g).to(dtype)
    assert torch.equal(o.concat_channels(a.to(DEV), b.to(DEV)).cpu(), torch.cat([a, b], -1))
    f = torch.randn(1001, generator=g)
    assert torch.equal(o.cast(f.to(DEV), torch.bfloat16).cpu(), f.to(torch.bfloat16))
    # timestep embedding vs the oracle
    from oracle.unet import timestep_embedding

    for t in (981.0, 1.0, 501.0):
        td = torch.tensor([t], device=DEV)
        got = o.timestep_embedding(td, 3, 320, torch.float32)
        ref = timestep_embedding(torch.full((3,), t), 320)
        assert max_err(got, ref) < 2e-4  # |arg| up to 1e3: a few ulp of the argument
    # row softmax
    s = torch.randn(70, 77, generator=g) * 4
    p = o.softmax_rows(F.pad(s, (0, 3)).contiguous().to(DEV), 77, 0.3, dtype, ldp=80)
    assert rel_err(p[:, :77].float(), torch.softmax(s * 0.3, -1)) < tol(dtype)
    assert float(p[:, 77:].float().abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------
# GEMM / conv
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 320), (8, 1280, 1280), (616, 320, 768), (4096, 320, 320),
                                   (2048, 2560, 320), (33, 4, 128), (70, 1000, 64), (1024, 1280, 5120)])
def test_gemm_nt(dtype, M, N, K):
    o = ops()
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dtype)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g).to(dtype)
    ref = a.double() @ w.double().t()
    got = o.gemm_nt(a.to(DEV), w.to(DEV))
    assert rel_err(got.float(), ref) < tol(dtype), "plain"
    got = o.gemm_nt(a.to(DEV), w.to(DEV), bias=bias.to(DEV), residual=res.to(DEV), alpha=0.5)
    assert rel_err(got.float(), 0.5 * ref + bias.double() + res.double()) < tol(dtype), "bias+residual"
    got = o.gemm_nt(a.to(DEV), w.to(DEV), bias=bias.to(DEV), act=o.ACT_SILU, out_dtype=torch.float32)
    assert got.dtype == torch.float32
    assert rel_err(got, F.silu(ref + bias.double())) < tol(dtype), "silu f32 out"
    rpg = 7
    rb = torch.randn((M + rpg - 1) // rpg, N, generator=g)
    got = o.gemm_nt(a.to(DEV), w.to(DEV), rowbias=rb.to(DEV), rows_per_group=rpg)
    assert rel_err(got.float(), ref + rb.double().repeat_interleave(rpg, 0)[:M]) < tol(dtype), "rowbias"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_gemm_batched_and_swapped(dtype):
    o = ops()
    g = torch.Generator().manual_seed(9)
    Bn, N, C = 3, 72, 128
    x = torch.randn(Bn, N, C, generator=g).to(dtype)
    wv = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dtype)
    ld = 80
    vt = o.gemm_nt(wv.to(DEV), x.to(DEV), ldc=ld)  # V^T[b] = W_v @ x[b]^T
    ref = torch.einsum("ck,bnk->bcn", wv.double(), x.double())
    assert vt.shape == (Bn, C, ld)
    assert rel_err(vt[:, :, :N].float(), ref) < tol(dtype)
    y = o.gemm_nt(x.to(DEV), torch.stack([wv, wv * 2, wv * 3]).to(DEV))
    ref2 = torch.stack([x[i].double() @ (wv.double() * (i + 1)).t() for i in range(3)])
    assert rel_err(y.float(), ref2) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,W,Cin,Cout,mode", [
    (2, 8, 8, 64, 64, "s1"), (1, 16, 12, 128, 320, "s1"), (2, 8, 8, 64, 128, "s2"), (1, 9, 7, 64, 64, "s2"),
    (2, 4, 6, 128, 64, "up"), (1, 8, 8, 64, 64, "pad1"), (1, 7, 9, 64, 64, "pad1"), (1, 64, 64, 320, 320, "s1"),
    (2, 8, 8, 64, 4, "s1"), (8, 8, 8, 1280, 1280, "s1"),
])
def test_conv3x3(dtype, B, H, W, Cin, Cout, mode):
    o = ops()
    g = torch.Generator().manual_seed(H * W + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g).to(dtype)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(dtype)
    bias = torch.randn(Cout, generator=g)
    xd, wd = x.double(), w.double()
    if mode == "s1":
        ref = F.conv2d(xd, wd, bias.double(), padding=1)
        kw = {}
    elif mode == "s2":
        ref = F.conv2d(xd, wd, bias.double(), stride=2, padding=1)
        kw = dict(stride=2)
    elif mode == "up":
        ref = F.conv2d(F.interpolate(xd, scale_factor=2.0, mode="nearest"), wd, bias.double(), padding=1)
        kw = dict(upsample=True)
    else:
        ref = F.conv2d(F.pad(xd, (0, 1, 0, 1)), wd, bias.double(), stride=2, padding=0)
        kw = dict(stride=2, pad_mode=1)
    Ho, Wo = ref.shape[-2:]
    xl = x.permute(0, 2, 3, 1).reshape(B, H * W, Cin).contiguous().to(DEV)
    wl = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(DEV)
    tb = torch.randn(B, Cout, generator=g)
    res = torch.randn(B, Ho * Wo, Cout, generator=g).to(dtype)
    y, ho, wo = o.conv3x3(xl, wl, B, H, W, bias=bias.to(DEV), rowbias=tb.to(DEV), residual=res.to(DEV), **kw)
    assert (ho, wo) == (Ho, Wo)
    full = ref + tb.double()[:, :, None, None]
    full = full.permute(0, 2, 3, 1).reshape(B, Ho * Wo, Cout) + res.double()
    assert rel_err(y.float(), full) < tol(dtype)
    y32, _, _ = o.conv3x3(xl, wl, B, H, W, bias=bias.to(DEV), out_dtype=torch.float32, **kw)
    assert y32.dtype == torch.float32
    assert rel_err(y32, ref.permute(0, 2, 3, 1).reshape(B, Ho * Wo, Cout)) < tol(dtype)


# ---------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, heads, scale):
    B, Nq, C = q.shape
    d = C // heads
    qh = q.double().view(B, Nq, heads, d).transpose(1, 2)
    kh = k.double().view(B, -1, heads, d).transpose(1, 2)
    vh = v.double().view(B, -1, heads, d).transpose(1, 2)
    p = torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1)
    return (p @ vh).transpose(1, 2).reshape(B, Nq, C)


@pytest.mark.parametrize("D", [32, 40, 64, 80, 160])
@pytest.mark.parametrize("Nq,Nk", [(256, 256), (64, 77), (16, 16), (200, 130), (1024, 1024)])
def test_attention_bf16(D, Nq, Nk):
    o = ops()
    heads, B = 2, 2
    C = heads * D
    g = torch.Generator().manual_seed(D * 7 + Nq + Nk)
    q = torch.randn(B, Nq, C, generator=g).bfloat16()
    k = torch.randn(B, Nk, C, generator=g).bfloat16()
    v = torch.randn(B, Nk, C, generator=g).bfloat16()
    ld = (Nk + 7) // 8 * 8
    vt = torch.full((B, C, ld), float("nan")).bfloat16()  # pad columns poisoned: the kernel must mask them
    vt[:, :, :Nk] = v.transpose(1, 2)
    scale = D ** -0.5
    got = o.attention(q.to(DEV), k.to(DEV), vt.to(DEV), heads, Nk, scale)
    ref = _attn_ref(q, k, v, heads, scale)
    assert torch.isfinite(got.float()).all()
    assert rel_err(got.float(), ref) < 1.2e-2
    assert max_err(got.float(), ref) < 6e-2


@pytest.mark.parametrize("spike", [1.5, 3.0, 8.0])
def test_attention_rescale_branch_spiked_keys(spike):
    """Force the running max to jump in a later tile: 1.5 -> the lagged-stabiliser rescale (d=40 kernel), 3.0 / 8.0 ->
    a jump of more than 2^20, which must take the classic-softmax fallback instead of overflowing."""
    o = ops()
    heads, B, D, N = 1, 1, 40, 320
    g = torch.Generator().manual_seed(77)
    q = torch.randn(B, N, D, generator=g)
    k = torch.randn(B, N, D, generator=g) * 0.3
    v = torch.randn(B, N, D, generator=g)
    k[0, 200] = q[0, 5] * spike   # query 5 sees a huge score at key 200 (4th tile)
    k[0, 310] = q[0, 100] * (spike + 1.0)
    q, k, v = q.bfloat16(), k.bfloat16(), v.bfloat16()
    vt = v.transpose(1, 2).contiguous()
    got = o.attention(q.to(DEV), k.to(DEV), vt.to(DEV), heads, N, D ** -0.5)
    ref = _attn_ref(q, k, v, heads, D ** -0.5)
    assert torch.isfinite(got.float()).all()
    assert max_err(got.float(), ref) < 6e-2 and rel_err(got.float(), ref) < 1.2e-2


@pytest.mark.parametrize("D", [40, 80])
def test_attention_large_and_negative_logits(D):
    """Scores far from zero in both directions: all-negative rows (the first-tile stabiliser must be the tile maximum, not
    zero) and logits of magnitude ~100."""
    o = ops()
    heads, B, N = 2, 1, 200
    g = torch.Generator().manual_seed(D)
    base = torch.randn(1, 1, heads * D, generator=g)
    q = (base * 4.0 + 0.3 * torch.randn(B, N, heads * D, generator=g))
    k = (-base * 4.0 + 0.3 * torch.randn(B, N, heads * D, generator=g))   # q.k ~ -16 |base|^2: every score << 0
    v = torch.randn(B, N, heads * D, generator=g)
    q, k, v = q.bfloat16(), k.bfloat16(), v.bfloat16()
    got = o.attention(q.to(DEV), k.to(DEV), v.transpose(1, 2).contiguous().to(DEV), heads, N, D ** -0.5)
    ref = _attn_ref(q, k, v, heads, D ** -0.5)
    assert torch.isfinite(got.float()).all()
    assert max_err(got.float(), ref) < 8e-2 and rel_err(got.float(), ref) < 3e-2


def test_attention_fused_qk_buffer():
    o = ops()
    heads, B, D, N = 8, 2, 40, 256
    C = heads * D
    g = torch.Generator().manual_seed(5)
    qk = torch.randn(B, N, 2 * C, generator=g).bfloat16()
    v = torch.randn(B, N, C, generator=g).bfloat16()
    got = o.attention(qk.to(DEV), qk.to(DEV), v.transpose(1, 2).contiguous().to(DEV), heads, N, D ** -0.5, k_col=C)
    ref = _attn_ref(qk[..., :C], qk[..., C:], v, heads, D ** -0.5)
    assert rel_err(got.float(), ref) < 1.2e-2


@pytest.mark.parametrize("dtype,d,heads", [(torch.float32, 40, 8), (torch.float32, 32, 2), (torch.bfloat16, 512, 1)])
def test_composed_attention(dtype, d, heads):
    from gm_diffusion.components.unet_2d_condition import composed_attention

    B, Nq, Nk = 2, 128, 77 if dtype == torch.float32 else 128
    C = heads * d
    g = torch.Generator().manual_seed(d)
    q = torch.randn(B, Nq, C, generator=g).to(dtype)
    k = torch.randn(B, Nk, C, generator=g).to(dtype)
    v = torch.randn(B, Nk, C, generator=g).to(dtype)
    mul = 64 if dtype == torch.bfloat16 else 4
    ld = (Nk + mul - 1) // mul * mul
    vt = torch.zeros(B, C, ld, dtype=dtype)
    vt[:, :, :Nk] = v.transpose(1, 2)
    got = composed_attention(q.to(DEV), 0, C, k.to(DEV), 0, C, vt.to(DEV), B, heads, d, Nq, Nk, d ** -0.5, dtype)
    ref = _attn_ref(q, k, v, heads, d ** -0.5)
    assert rel_err(got.float(), ref) < (2e-5 if dtype == torch.float32 else 1.5e-2)


@pytest.mark.parametrize("M,C", [(300, 320), (4096, 640), (64, 1280), (33, 64)])
def test_gemm_fused_geglu_epilogue(M, C):
    """ff.net.0 (GEGLU) with the gate applied in the GEMM epilogue: rows of W interleaved [16 value | 16 gate]."""
    o = ops()
    g = torch.Generator().manual_seed(C + M)
    x = torch.randn(M, C, generator=g).bfloat16()
    w = (torch.randn(8 * C, C, generator=g) / math.sqrt(C)).bfloat16()
    b = torch.randn(8 * C, generator=g)
    half = 4 * C
    wi = torch.stack([w[:half].reshape(half // 16, 16, -1), w[half:].reshape(half // 16, 16, -1)], 1).reshape(2 * half, -1).contiguous()
    bi = torch.stack([b[:half].reshape(half // 16, 16), b[half:].reshape(half // 16, 16)], 1).reshape(2 * half).contiguous()
    got = o.gemm_nt(x.to(DEV), wi.to(DEV), bias=bi.to(DEV), act=o.ACT_GEGLU)
    assert got.shape == (M, half)
    y = x.double() @ w.double().t() + b.double()
    ref = y[:, :half] * F.gelu(y[:, half:])
    assert rel_err(got.float(), ref) < 1.5e-2
    unfused = o.geglu(o.gemm_nt(x.to(DEV), w.to(DEV), bias=b.to(DEV)))
    assert rel_err(got.float(), unfused.float()) < 1.5e-2


def test_rgbe_encode_bit_exact_and_file(tmp_path):
    from gm_diffusion import hdr
    from oracle import hdr_ops as H

    o = ops()
    g = torch.Generator().manual_seed(12)
    x = torch.rand(37, 53, 3, generator=g) * 30 - 0.2
    x[0, 0] = 0.0
    x[0, 1] = torch.tensor([1e-35, 0.0, 0.0])
    got = o.rgbe_encode(x.to(DEV)).cpu().numpy()
    assert np.array_equal(got, H.rgbe_encode(x.numpy()))  # byte work: bit exact
    path = tmp_path / "a.hdr"
    hdr.save_hdr_image(x.to(DEV), str(path), compression="none")
    raw = open(path, "rb").read()
    head = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 37 +X 53\n"
    assert raw.startswith(head) and len(raw) == len(head) + 37 * 53 * 4
    assert np.array_equal(np.frombuffer(raw[len(head):], np.uint8).reshape(37, 53, 4), got)
    # default: run-length framed scanlines (what cv2.imwrite writes): the oracle's restatement byte for byte, and readable back
    hdr.save_hdr_image(x.to(DEV), str(path))
    raw = open(path, "rb").read()
    assert raw.startswith(head) and raw[len(head):] == H.rgbe_rle_scanlines(got.reshape(37, 53, 4))
    assert np.array_equal(H.rgbe_rle_decode(raw[len(head):], 37, 53), got.reshape(37, 53, 4))


@pytest.mark.parametrize("D,N", [(64, 77), (32, 77), (64, 200)])
def test_attention_causal(D, N):
    """Causal mask of the CLIP text encoder (gmd_attention causal=1), incl. a ragged last tile and multi-tile rows."""
    o = ops()
    heads, B = 3, 2
    C = heads * D
    g = torch.Generator().manual_seed(D + N)
    q = torch.randn(B, N, C, generator=g).bfloat16()
    k = torch.randn(B, N, C, generator=g).bfloat16()
    v = torch.randn(B, N, C, generator=g).bfloat16()
    ld = (N + 7) // 8 * 8
    vt = torch.full((B, C, ld), float("nan")).bfloat16()
    vt[:, :, :N] = v.transpose(1, 2)
    got = o.attention(q.to(DEV), k.to(DEV), vt.to(DEV), heads, N, D ** -0.5, causal=True)
    qf, kf, vf = [t.float().view(B, N, heads, D).transpose(1, 2) for t in (q, k, v)]
    s = qf @ kf.transpose(-1, -2) * D ** -0.5 + torch.full((N, N), float("-inf")).triu(1)
    ref = (torch.softmax(s, -1) @ vf).transpose(1, 2).reshape(B, N, C)
    assert torch.isfinite(got.float()).all()
    assert rel_err(got.float(), ref) < 1.2e-2 and max_err(got.float(), ref) < 6e-2


def test_attention_causal_unsupported_head_dim_fails_loudly():
    o = ops()
    q = torch.zeros(1, 16, 40, dtype=torch.bfloat16, device=DEV)
    vt = torch.zeros(1, 40, 16, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(Exception, match="causal"):
        o.attention(q, q, vt, 1, 16, 1.0, causal=True)


def test_softmax_rows_causal_embedding_and_quick_gelu():
    o = ops()
    g = torch.Generator().manual_seed(3)
    s = torch.randn(2, 5, 8, generator=g)           # rows = (head, query) with Nq = 5
    p = o.softmax_rows(s.to(DEV), 5, 0.7, torch.float32, ldp=8, causal_nq=5).cpu()
    m = (s[..., :5] * 0.7) + torch.full((5, 5), float("-inf")).triu(1)
    assert torch.allclose(p[..., :5], torch.softmax(m, -1), atol=1e-6) and float(p[..., 5:].abs().max()) == 0.0
    table, pos = torch.randn(50, 16, generator=g), torch.randn(7, 16, generator=g)
    ids = torch.randint(0, 50, (3, 7), generator=g)
    for dt in (torch.float32, torch.bfloat16):
        e = o.embedding_lookup(ids.to(DEV), table.to(DEV, dt), pos.to(DEV, dt)).float().cpu()
        assert torch.allclose(e, table.to(dt).float()[ids] + pos.to(dt).float()[None], atol=2e-2 if dt == torch.bfloat16 else 0)
    x = torch.randn(70, 64, generator=g)
    w = torch.randn(96, 64, generator=g) / 8
    b = torch.randn(96, generator=g)
    z = x @ w.T + b
    ref = z * torch.sigmoid(1.702 * z)
    got = o.gemm_nt(x.to(DEV), w.to(DEV), bias=b.to(DEV), act=o.ACT_QUICK_GELU).cpu()
    assert rel_err(got, ref) < 1e-5
    gb = o.gemm_nt(x.bfloat16().to(DEV), w.bfloat16().to(DEV), bias=b.to(DEV), act=o.ACT_QUICK_GELU).float().cpu()
    assert rel_err(gb, ref) < 2e-2


@pytest.mark.parametrize("do_cfg,gr", [(True, 0.0), (True, 0.7), (False, 0.0)])
def test_dpm_step_kernel_bit_exact_vs_torch(do_cfg, gr):
    """gmd_dpm_step against the torch expressions of the host path (= diffusers' operation order), over a whole 9-step
    DPM-Solver++ trajectory: first-order start, second-order middle, lower-order final step with sigma_last = 0."""
    from gm_diffusion.components import DPMSolverMultistepScheduler

    mk = lambda: DPMSolverMultistepScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1,
                                             timestep_spacing="leading")
    host, dev = mk(), mk()
    host.set_timesteps(9)
    dev.set_timesteps(9)
    g = torch.Generator().manual_seed(17)
    B, gs = 2, 7.5
    x = torch.randn(B, 4, 8, 8, generator=g)
    xd = x.to(DEV)
    for i, t in enumerate(host.timesteps):
        raw = torch.randn((2 * B if do_cfg else B), 4, 8, 8, generator=g)
        if do_cfg:
            u, c = raw.chunk(2)
            eps = u + gs * (c - u)
            if gr > 0:
                std_t = c.std(dim=list(range(1, c.ndim)), keepdim=True)
                std_c = eps.std(dim=list(range(1, eps.ndim)), keepdim=True)
                eps = gr * (eps * (std_t / std_c)) + (1 - gr) * eps
        else:
            eps = raw
        a = host.alphas_cumprod[int(t)]
        x0_ref = (x - (1 - a).sqrt() * eps) / a.sqrt()
        x = host.step(eps, t, x, return_dict=False)[0]
        xd, x0 = dev.fused_step(raw.to(DEV), int(t), xd, do_cfg, gs, gr, want_x0=True)
        if gr == 0.0:
            assert torch.equal(xd.cpu(), x), (i, float((xd.cpu() - x).abs().max()))
            assert torch.equal(x0.cpu(), x0_ref)
        else:  # the std ratio is reduced in a different order on the device
            assert torch.allclose(xd.cpu(), x, atol=2e-5) and torch.allclose(x0.cpu(), x0_ref, atol=2e-5)
    assert torch.isfinite(xd).all()


def test_empty_inputs_are_no_ops():
    """Zero-sized batches go through every launcher as no-ops (empty result, no launch, no error)."""
    o = ops()
    bf = torch.bfloat16
    e = lambda *s: torch.empty(*s, dtype=bf, device=DEV)
    w = torch.zeros(64, 64, dtype=bf, device=DEV)
    assert o.gemm_nt(e(0, 64), w).shape == (0, 64)
    y, ho, wo = o.conv3x3(e(0, 64, 64), torch.zeros(64, 9 * 64, dtype=bf, device=DEV), 0, 8, 8)
    assert y.shape == (0, 64, 64) and (ho, wo) == (8, 8)
    assert o.attention(e(0, 16, 64), e(0, 16, 64), e(0, 64, 16), 2, 16, 1.0).shape == (0, 16, 64)
    assert o.layernorm(e(0, 64), torch.ones(64, device=DEV), torch.zeros(64, device=DEV)).shape == (0, 64)
    assert o.cast(e(0, 3), torch.float32).numel() == 0
    z = torch.empty(0, 4, 8, 8, device=DEV)
    eps, xp, x0 = o.latent_step(z, z, 0, (1.0, 0.5, 1.0, 1.0, 0.0), False, 1.0, want_x0=True)
    assert xp.numel() == 0 and x0.numel() == 0


def test_gemm_conv_epilogue_fuzz():
    """Seeded sweep over shapes that straddle every epilogue path of the bf16 kernels (row-contiguous LDS epilogue on full
    tiles, register epilogue on ragged edges / float32 outputs, split-K slabs, 64x64 kernel, batched launches) with all
    combinations of bias / row bias (group sizes around the 64-row wave tile) / residual / activation / alpha."""
    o = ops()
    rng = np.random.default_rng(2024)
    g = torch.Generator().manual_seed(2024)
    acts = [(o.ACT_NONE, lambda z: z), (o.ACT_SILU, F.silu), (o.ACT_QUICK_GELU, lambda z: z * torch.sigmoid(1.702 * z))]
    cases = 0
    for _ in range(36):
        M = int(rng.choice([128, 256, 384, 640, 1000, 1024, 2048, 4096, 16384]))
        N = int(rng.choice([64, 128, 160, 320, 328, 640, 1280]))
        K = int(rng.choice([64, 128, 320, 640, 1536, 2560]))
        if M * N * K > 16384 * 640 * 640:
            continue
        use_bias, use_rb, use_res = (bool(v) for v in rng.integers(0, 2, 3))
        act, fn = acts[int(rng.integers(0, 3))]
        alpha = float(rng.choice([1.0, 0.5]))
        out_f32 = bool(rng.integers(0, 4) == 0) and not use_res
        a = torch.randn(M, K, generator=g).bfloat16()
        w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16()
        ref = alpha * (a.double() @ w.double().t())
        kw = {}
        if use_bias:
            b = torch.randn(N, generator=g)
            kw["bias"] = b.to(DEV)
            ref = ref + b.double()
        if use_rb:
            rpg = int(rng.choice([16, 48, 64, 100, 256, M]))
            rb = torch.randn((M + rpg - 1) // rpg, N, generator=g)
            kw.update(rowbias=rb.to(DEV), rows_per_group=rpg)
            ref = ref + rb.double().repeat_interleave(rpg, 0)[:M]
        if use_res:
            r = torch.randn(M, N, generator=g).bfloat16()
            kw["residual"] = r.to(DEV)
            ref = ref + r.double()
        got = o.gemm_nt(a.to(DEV), w.to(DEV), alpha=alpha, act=act, out_dtype=torch.float32 if out_f32 else None, **kw)
        assert got.shape == (M, N)
        assert rel_err(got.float(), fn(ref)) < 1.2e-2, (M, N, K, use_bias, use_rb, use_res, act, alpha, out_f32)
        cases += 1
    for _ in range(14):  # conv3x3: spatial sizes around the tile, stride / upsample, row bias = per-sample time embedding
        B = int(rng.choice([1, 2, 3, 8]))
        H = int(rng.choice([8, 12, 16, 32, 64]))
        W = int(rng.choice([8, 16, 20, 32, 64]))
        ci, co = int(rng.choice([64, 128, 320])), int(rng.choice([64, 160, 320]))
        mode = int(rng.integers(0, 3))  # 0 plain, 1 stride 2, 2 upsample
        x = torch.randn(B, ci, H, W, generator=g).bfloat16()
        wt = (torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci)).bfloat16()
        b = torch.randn(co, generator=g)
        tb = torch.randn(B, co, generator=g)
        xin = F.interpolate(x.double(), scale_factor=2, mode="nearest") if mode == 2 else x.double()
        ref = F.conv2d(xin, wt.double(), b.double(), stride=2 if mode == 1 else 1, padding=1) + tb.double()[:, :, None, None]
        xl = x.permute(0, 2, 3, 1).reshape(B, H * W, ci).contiguous().to(DEV)
        wl = wt.permute(0, 2, 3, 1).reshape(co, 9 * ci).contiguous().to(DEV)
        y, ho, wo = o.conv3x3(xl, wl, B, H, W, bias=b.to(DEV), rowbias=tb.to(DEV), stride=2 if mode == 1 else 1, upsample=mode == 2)
        assert (ho, wo) == tuple(ref.shape[-2:])
        got = y.float().cpu().view(B, ho, wo, co).permute(0, 3, 1, 2)
        assert rel_err(got, ref) < 1.2e-2, (B, H, W, ci, co, mode)
        cases += 1
    assert cases >= 40


def test_attention_and_norm_fuzz():
    """Seeded sweep: attention over head dims / ragged token counts / fused-QK buffers, GroupNorm over both kernels' ranges
    (fused with 16/8/4-byte rows, two-launch split), LayerNorm over the one- / two- / four-row variants."""
    o = ops()
    rng = np.random.default_rng(7)
    g = torch.Generator().manual_seed(7)
    for _ in range(14):
        D = int(rng.choice([32, 40, 64, 80, 160]))
        H = int(rng.choice([1, 2, 8]))
        B = int(rng.choice([1, 2, 3]))
        Nq = int(rng.choice([1, 31, 64, 130, 256, 777]))
        self_attn = bool(rng.integers(0, 2))
        Nk = Nq if self_attn else int(rng.choice([1, 7, 77, 154, 300]))
        C = H * D
        k = torch.randn(B, Nk, C, generator=g).bfloat16()
        v = torch.randn(B, Nk, C, generator=g).bfloat16()
        ld = (Nk + 7) // 8 * 8
        vt = torch.full((B, C, ld), float("nan")).bfloat16()
        vt[:, :, :Nk] = v.transpose(1, 2)
        if self_attn:  # q and k as two column blocks of one fused projection buffer
            qk = torch.randn(B, Nq, 2 * C, generator=g).bfloat16()
            qk[..., C:] = k
            q = qk[..., :C]
            got = o.attention(qk.to(DEV), qk.to(DEV), vt.to(DEV), H, Nk, D ** -0.5, k_col=C)
        else:
            q = torch.randn(B, Nq, C, generator=g).bfloat16()
            got = o.attention(q.to(DEV), k.to(DEV), vt.to(DEV), H, Nk, D ** -0.5)
        ref = _attn_ref(q, k, v, H, D ** -0.5)
        assert torch.isfinite(got.float()).all()
        assert rel_err(got.float(), ref) < 1.3e-2, (D, H, B, Nq, Nk, self_attn)
    for _ in range(12):
        B = int(rng.choice([1, 2, 4]))
        HW = int(rng.choice([16, 64, 100, 256, 1024, 4096]))
        C = int(rng.choice([64, 320, 640, 960, 1280, 1920]))
        silu = bool(rng.integers(0, 2))
        x = (torch.randn(B, HW, C, generator=g) * 2 + 0.5).bfloat16()
        ga, be = torch.randn(C, generator=g), torch.randn(C, generator=g)
        y = F.group_norm(x.float().transpose(1, 2), 32, ga, be, 1e-5)
        y = (F.silu(y) if silu else y).transpose(1, 2)
        got = o.groupnorm(x.to(DEV), B, 32, ga.to(DEV), be.to(DEV), 1e-5, silu=silu)
        assert rel_err(got.float(), y) < 1e-2, ("gn", B, HW, C, silu)
    for rows, C in [(5, 64), (300, 320), (8192, 320), (9000, 320), (4096, 640), (4100, 1024), (70, 1280), (8192, 512)]:
        x = (torch.randn(rows, C, generator=g) * 3 - 1).bfloat16()
        ga, be = torch.randn(C, generator=g), torch.randn(C, generator=g)
        got = o.layernorm(x.to(DEV), ga.to(DEV), be.to(DEV), 1e-5)
        assert rel_err(got.float(), F.layer_norm(x.float(), (C,), ga, be, 1e-5)) < 1e-2, ("ln", rows, C)


def test_plan_override_cannot_bypass_geglu_tile_rule():
    """gmd_gemm_plan_override is a tuning hook; a forced tile whose kernel has no GEGLU epilogue must be refused, not
    launched (the plain epilogue would write [M, N] into the [M, N/2] output)."""
    from gm_diffusion._native import HipExtensionError, lib

    o = ops()
    g = torch.Generator().manual_seed(2)
    a = torch.randn(256, 320, generator=g).bfloat16().to(DEV)
    w = (torch.randn(640, 320, generator=g) * 0.05).bfloat16().to(DEV)
    b = torch.randn(640, generator=g).to(DEV)
    want = o.gemm_nt(a, w, bias=b, act=o.ACT_GEGLU)
    had = os.environ.pop("GMD_TUNING", None)
    assert lib().gmd_gemm_plan_override(128, 160, 9, 1) != 0, "overrides are refused outside a GMD_TUNING=1 process"
    assert torch.equal(o.gemm_nt(a, w, bias=b, act=o.ACT_GEGLU), want)  # ... and the refused override left the heuristic in place
    os.environ["GMD_TUNING"] = "1"
    try:
        # the plans of the round-2 fault record (tools/bench_graph_ops.py under 64,64,103 / 64,64,104 / 128,160,123 / 128,160,124:
        # odd number of 16-column tiles per wave, no GEGLU epilogue) and the default-pipeline 128x160 tile
        for plan in ((128, 160, 9, 1), (64, 64, 103, 0), (64, 64, 104, 0), (128, 160, 123, 0), (128, 160, 124, 0), (128, 128, 9, 2)):
            assert lib().gmd_gemm_plan_override(*plan) == 0
            with pytest.raises(HipExtensionError):
                o.gemm_nt(a, w, bias=b, act=o.ACT_GEGLU)
        lib().gmd_gemm_plan_override(128, 128, 9, 1)  # an even-tile kernel: allowed, same result as the heuristic's choice
        got = o.gemm_nt(a, w, bias=b, act=o.ACT_GEGLU)
    finally:
        lib().gmd_gemm_plan_override(0, 0, 0, 0)
        if had is None:
            os.environ.pop("GMD_TUNING", None)
        else:
            os.environ["GMD_TUNING"] = had
    assert rel_err(got.float(), want.float()) < 1e-2


# ---------------------------------------------------------------------------------------------
# GroupNorm statistics out of the producer's epilogue
# ---------------------------------------------------------------------------------------------
def _bucket_sums(y, bucket):
    """float64 {sum, sumsq} of the stored values per 64-row block and per bucket of adjacent columns."""
    M, N = y.shape
    v = y.double().cpu().view(M // 64, 64, N // bucket, bucket)
    return torch.stack([v.sum((1, 3)), (v * v).sum((1, 3))], -1)


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("kind", ["conv_res", "conv_rowbias", "conv_up", "gemm_res"])
def test_colstats_from_producer_epilogue(dtype, kind):
    """``colstats=True``: the launch that writes a tensor also leaves the per-(64 rows x 10 channels) sums of the values it
    STORED -- compared with float64 sums of the stored tensor itself, so rounding is not part of the error."""
    o = ops()
    g = torch.Generator().manual_seed(5)
    if kind == "gemm_res":
        M, N, K = 16384, 320, 320
        a = (torch.randn(M, K, generator=g) * 0.5).to(DEV, dtype)
        w = (torch.randn(N, K, generator=g) * 0.05).to(DEV, dtype)
        res = torch.randn(M, N, generator=g).to(DEV, dtype)
        bias = torch.randn(N, generator=g).to(DEV)
        y = o.gemm_nt(a, w, bias=bias, residual=res, colstats=True)
        plain = o.gemm_nt(a, w, bias=bias, residual=res)
    else:
        B, H, W, cin, cout = 4, 64, 64, 64, 320
        if kind == "conv_up":
            H = W = 32
        x = torch.randn(B, H * W, cin, generator=g).to(DEV, dtype)
        w = (torch.randn(cout, 9 * cin, generator=g) * 0.04).to(DEV, dtype)
        bias = torch.randn(cout, generator=g).to(DEV)
        kw = {}
        if kind == "conv_res":
            kw["residual"] = torch.randn(B, 64 * 64, cout, generator=g).to(DEV, dtype)
        if kind == "conv_rowbias":
            kw["rowbias"] = torch.randn(B, cout, generator=g).to(DEV)
        if kind == "conv_up":
            kw["upsample"] = True
        y, _, _ = o.conv3x3(x, w, B, H, W, bias=bias, colstats=True, **kw)
        plain, _, _ = o.conv3x3(x, w, B, H, W, bias=bias, **kw)
    assert torch.equal(y, plain)  # the statistics do not touch the output
    st, C = y._colstats
    assert C == y.shape[-1] and st.dtype == torch.float32
    ref = _bucket_sums(y.view(-1, C), o.COLSTATS_BUCKET)
    assert tuple(st.shape) == tuple(ref.shape)
    assert rel_err(st, ref) < 2e-6 and max_err(st[..., 0], ref[..., 0]) < 2e-3


@pytest.mark.parametrize("B,H,ci,co", [(2, 64, 320, 320), (4, 32, 640, 640), (4, 32, 1280, 640)])
def test_split_launch_with_in_kernel_reduction_emits_statistics(B, H, ci, co):
    """Round 5: under the co-running plan family launches of fewer than ~128 tiles of 256 rows run as 2 to 4 K slices (the GM UNet's
    32x32 convolutions, both UNets' 16x16 level); the last slice adds the others' fragments inside the kernel and leaves through the full-tile row epilogue, so the
    launch emits the GroupNorm statistics like an unsplit one -- sums of the STORED values against float64, output untouched."""
    o = ops()
    g = torch.Generator().manual_seed(B + H + ci)
    x = torch.randn(B, H * H, ci, generator=g).bfloat16().to(DEV)
    w = (torch.randn(co, 9 * ci, generator=g) * 0.02).bfloat16().to(DEV)
    bias = torch.randn(co, generator=g).to(DEV)
    rb = torch.randn(B, co, generator=g).to(DEV)
    with o.plan_family(1):
        assert o.gemm_plan_info(torch.bfloat16, B * H * H, co, 9 * ci)[3] in (2, 3, 4)  # (the statistics launch itself keeps 160-column tiles: 2 or 4)
        y, _, _ = o.conv3x3(x, w, B, H, H, bias=bias, rowbias=rb, colstats=True)
        plain, _, _ = o.conv3x3(x, w, B, H, H, bias=bias, rowbias=rb)
    # (the statistics launch keeps 160-column tiles, the plain one may take 128-column tiles with another number of K slices: the
    #  same product up to float32 summation order)
    assert hasattr(y, "_colstats") and rel_err(y.float(), plain.float()) < 4e-3
    st, C = y._colstats
    ref = _bucket_sums(y.view(-1, C), o.COLSTATS_BUCKET)
    assert tuple(st.shape) == tuple(ref.shape) and rel_err(st, ref) < 2e-6 and max_err(st[..., 0], ref[..., 0]) < 2e-3
    # ... and a GroupNorm fed from them equals the GroupNorm that computes its own
    gamma, beta = torch.randn(co, generator=g).to(DEV), torch.randn(co, generator=g).to(DEV)
    n_cs = o.groupnorm(y, B, 32, gamma, beta, 1e-5, silu=True)
    n_own = o.groupnorm(plain, B, 32, gamma, beta, 1e-5, silu=True)
    assert rel_err(n_cs.float(), n_own.float()) < 4e-3


def test_colstats_only_where_the_plan_has_the_row_epilogue():
    """Small / split-K / 64x64-tile launches cannot emit statistics: the front end then attaches none (GroupNorm falls back
    to its own statistics launch), and forcing the request through the C ABI fails loudly instead of leaving them unwritten."""
    from gm_diffusion import _native

    o = ops()
    lib = _native.lib()
    code = o.dtype_code(torch.bfloat16)
    assert lib.gmd_gemm_colstats_plan(code, 32768, 320, 2880, 1, o.WORKSPACE_BYTES, 10) == 1
    # split-K: since round 5 launches of up to four K slices reduce inside the kernel and their last slice runs the row epilogue,
    # statistics included (test_split_launch_with_in_kernel_reduction_emits_statistics); the slab path (more slices, or the in-kernel
    # reduction switched off) still cannot
    prev = lib.gmd_splitk_fixup_max(0)
    try:
        assert lib.gmd_gemm_colstats_plan(code, 2048, 1280, 11520, 1, o.WORKSPACE_BYTES, 10) == 0
    finally:
        lib.gmd_splitk_fixup_max(prev)
    with o.plan_family(1):
        assert lib.gmd_gemm_colstats_plan(code, 256, 1280, 5120, 1, o.WORKSPACE_BYTES, 10) == 0  # 8 slices: slabs
    assert lib.gmd_gemm_colstats_plan(code, 512, 320, 320, 1, o.WORKSPACE_BYTES, 10) == 0      # 64x64 tiles
    assert lib.gmd_gemm_colstats_plan(code, 32768, 320, 2880, 1, o.WORKSPACE_BYTES, 32) == 0   # 80 % 32 != 0
    assert lib.gmd_gemm_colstats_plan(o.dtype_code(torch.float32), 32768, 320, 2880, 1, o.WORKSPACE_BYTES, 10) == 0
    a = torch.randn(512, 320, device=DEV).bfloat16()
    w = torch.randn(320, 320, device=DEV).bfloat16()
    assert not hasattr(o.gemm_nt(a, w, colstats=True), "_colstats")
    out = torch.empty(512, 320, device=DEV, dtype=torch.bfloat16)
    st = torch.zeros(8, 32, 2, device=DEV)
    rc = lib.gmd_gemm_nt(a.data_ptr(), w.data_ptr(), out.data_ptr(), code, code, 512, 320, 320, 320, 320, 320, 1, 0, 0, 0, None, None, 0,
                         0, None, 0, 0, 1.0, 0, st.data_ptr(), 10, None, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 3 and b"column statistics" in lib.gmd_last_error()


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("Ca,Cb,silu", [(320, 0, True), (640, 320, True), (320, 320, False), (1280, 640, True)])
def test_groupnorm_from_producer_colstats(dtype, Ca, Cb, silu):
    """GroupNorm fed by producer statistics (one producer, or the two halves of a skip concatenation whose groups straddle
    the seam: 960 / 32 = 30 channels per group over a 640 | 320 split) against float64 GroupNorm of the same stored tensor
    and against the kernel's own statistics path."""
    o = ops()
    g = torch.Generator().manual_seed(Ca + Cb)
    B, HW, G = 2, 1024, 32
    C = Ca + Cb

    def produce(c, seed):
        gg = torch.Generator().manual_seed(seed)
        a = (torch.randn(B * HW, 320, generator=gg) * 0.5).to(DEV, dtype)
        w = (torch.randn(c, 320, generator=gg) * 0.08).to(DEV, dtype)
        bias = (torch.randn(c, generator=gg) * 2.0).to(DEV)  # non-zero means: exercises the E[x^2] - E[x]^2 form
        # M = 2048 is too small for the ring kernel alone: run the producer on 8 stacked copies and keep the first
        big = o.gemm_nt(a.repeat(8, 1), w, bias=bias, colstats=True)
        st, _ = big._colstats
        y = big[: B * HW].contiguous().view(B, HW, c)
        y._colstats = (st[: B * HW // 64].contiguous(), c)
        return y

    ya = produce(Ca, 1)
    x = ya
    if Cb:
        x = o.concat_channels(ya, produce(Cb, 2))
        assert isinstance(x._colstats, list)
    gamma, beta = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
    before = o.colstats_uses
    got = o.groupnorm(x, B, G, gamma, beta, 1e-5, silu=silu)
    assert o.colstats_uses == before + 1
    xs = x.clone()  # a fresh tensor carries no statistics: the kernel's own two-launch path
    assert not hasattr(xs, "_colstats")
    own = o.groupnorm(xs, B, G, gamma, beta, 1e-5, silu=silu)
    assert o.colstats_uses == before + 1
    ref = F.group_norm(x.double().cpu().permute(0, 2, 1), G, gamma.double().cpu(), beta.double().cpu(), 1e-5).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    assert rel_err(got, ref) < tol(dtype) and rel_err(got, own) < 1e-3


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("B,Nq,Nk,H", [(4, 2048, 2048, 8), (4, 2048, 77, 8), (3, 1000, 130, 4)])
def test_attention_d40_dma_ring_is_repeatable_beside_another_stream(dtype, B, Nq, Nk, H):
    """The d = 40 kernel stages K / V^T with LDS-DMA behind counted waits: an ordering bug there shows up as an occasional
    wrong tile, not as a steady error.  Same inputs, many launches, a second stream keeping the memory system busy: every output
    must be bit-identical to the first (the steady-state accuracy is covered by test_attention_bf16 / test_attention_f16)."""
    o = ops()
    g = torch.Generator().manual_seed(17)
    C = H * 40
    q = torch.randn(B, Nq, C, generator=g).to(DEV, dtype)
    k = torch.randn(B, Nk, C, generator=g).to(DEV, dtype)
    vt = torch.randn(B, C, (Nk + 7) // 8 * 8, generator=g).to(DEV, dtype)
    xa = torch.randn(8192, 320, generator=g).to(DEV, dtype)
    wa = torch.randn(320, 320, generator=g).to(DEV, dtype)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    first = o.attention(q, k, vt, H, Nk, 40 ** -0.5).clone()
    differ = 0
    for it in range(40):
        if it % 4 == 0:
            with torch.cuda.stream(side):
                for _ in range(3):
                    o.gemm_nt(xa, wa)
        differ += int(not torch.equal(o.attention(q, k, vt, H, Nk, 40 ** -0.5), first))
    torch.cuda.synchronize()
    assert differ == 0


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("D,Nk", [(40, 77), (40, 203), (80, 77), (40, 130)])
def test_attention_partial_last_tile_never_reads_poisoned_memory(dtype, D, Nk):
    """Partial last key tile (Nk % 64 != 0, Nk % 8 != 0): K rows at and beyond Nk exist in memory and are NaN, the V^T row
    padding beyond Nk is NaN -- the kernels' range-checked loads / masking / LDS zeroing must keep every NaN out (the d = 40
    kernel stages tiles by LDS-DMA with hand-counted waits: csrc/attention.hip, attn40_kernel)."""
    o = ops()
    H, B, Nq = 8, 2, 192
    C = H * D
    g = torch.Generator().manual_seed(Nk + D)
    q = torch.randn(B, Nq, C, generator=g).to(dtype)
    kfull = torch.full((B, Nk + 70, C), float("nan")).to(dtype)  # rows >= Nk: poison
    k = torch.randn(B, Nk, C, generator=g).to(dtype)
    kfull[:, :Nk] = k
    v = torch.randn(B, Nk, C, generator=g).to(dtype)
    ld = (Nk + 7) // 8 * 8 + 16
    vt = torch.full((B, C, ld), float("nan")).to(dtype)
    vt[:, :, :Nk] = v.transpose(1, 2)
    ref = _attn_ref(q.float(), k.float(), v.float(), H, D ** -0.5)
    got = o.attention(q.to(DEV), kfull.to(DEV), vt.to(DEV), H, Nk, D ** -0.5)
    assert torch.isfinite(got.float()).all()
    assert rel_err(got.float(), ref) < (2e-2 if dtype == torch.bfloat16 else 3e-3)


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("M", [128, 4096, 16384])
def test_ff_geglu_fused_matches_two_gemms_and_float64(dtype, M):
    """csrc/ff_fused.hip: the whole GEGLU feed-forward of a level-0 transformer block in one launch (the [tokens, 4C] tensor
    stays on chip) against the two-launch path it replaces (same operands, same dtype) and a float64 reference."""
    o = ops()
    C = 320
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, C, generator=g).to(dtype)
    res = torch.randn(M, C, generator=g).to(dtype)
    wf = (torch.randn(8 * C, C, generator=g) / math.sqrt(C)).to(dtype)
    bf = torch.randn(8 * C, generator=g) * 0.2
    w2 = (torch.randn(C, 4 * C, generator=g) / math.sqrt(4 * C)).to(dtype)
    b2 = torch.randn(C, generator=g) * 0.2
    half = 4 * C
    wi = torch.stack([wf[:half].reshape(half // 16, 16, -1), wf[half:].reshape(half // 16, 16, -1)], 1).reshape(2 * half, -1).contiguous()
    bi = torch.stack([bf[:half].reshape(half // 16, 16), bf[half:].reshape(half // 16, 16)], 1).reshape(2 * half).contiguous()
    z = x.double() @ wf.double().t() + bf.double()
    hmid = z[:, :half] * F.gelu(z[:, half:])
    ref = hmid @ w2.double().t() + b2.double() + res.double()
    xd, rd, wid_, bid, w2d, b2d = x.to(DEV), res.to(DEV), wi.to(DEV), bi.to(DEV), w2.to(DEV), b2.to(DEV)
    assert o.ff_fused_ok(xd, C, min_rows=0)
    got = o.ff_geglu_fused(xd, wid_, bid, w2d, b2d, rd)
    two = o.gemm_nt(o.gemm_nt(xd, wid_, bias=bid, act=o.ACT_GEGLU), w2d, bias=b2d, residual=rd)
    t = 1.5e-2 if dtype == torch.bfloat16 else 2e-3
    assert rel_err(got.float(), ref) < t and rel_err(two.float(), ref) < t
    assert rel_err(got.float(), two.float()) < t  # both round H to the 16-bit type once; only the summation order differs
    assert not o.ff_fused_ok(xd[:100], C, min_rows=0) and not o.ff_fused_ok(torch.zeros(128, 640, dtype=dtype, device=DEV), 640, min_rows=0)
    assert o.ff_fused_ok(xd, C) == (M >= o.FUSED_FF_MIN_ROWS)  # the product path takes it only where it fills the chip
    with o.plan_family(1):  # ... and, inside a co-running forward, from half the chip up (CU-time beside the other stream's kernels)
        assert o.ff_fused_ok(xd, C) == (M >= o.FUSED_FF_MIN_ROWS_CO_RUN)


# B, H, Cin, Cout, split-K expected, accesses per thread class
CONV_GN_CASES = [(2, 16, 320, 256), (8, 16, 640, 1280), (8, 8, 1280, 1280), (2, 16, 320, 2560), (1, 8, 640, 1280), (16, 8, 320, 2560)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("B,H,Cin,Cout", CONV_GN_CASES)
def test_conv3x3_groupnorm_from_splitk_slabs_bit_identical(dtype, B, H, Cin, Cout):
    """gmd_conv3x3_groupnorm (ResnetBlock2D conv1 -> + time embedding -> norm2 -> SiLU on the 16x16 / 8x8 levels): the GroupNorm
    kernel sums the split-K slabs itself.  Raw and normalised tensors must equal gmd_conv3x3 + gmd_groupnorm_fused bit for bit,
    with and without bias / row bias (column offset into a wider matrix) / residual / SiLU; where the launch does not fuse the
    wrapper falls back to the two calls."""
    from gm_diffusion import hip_ops as ops
    from gm_diffusion._native import lib
    g = torch.Generator().manual_seed(B * 1000 + H + Cin + Cout)
    W = H
    x = torch.randn(B, H * W, Cin, generator=g).to(DEV, dtype)
    w = (torch.randn(Cout, 9 * Cin, generator=g) / math.sqrt(9 * Cin)).to(DEV, dtype)
    if dtype == torch.float32:
        w = ops.split_weights(w)
    bias = torch.randn(Cout, generator=g).to(DEV)
    temb = torch.randn(B, 3 * Cout, generator=g).to(DEV)
    res = torch.randn(B, H * W, Cout, generator=g).to(DEV, dtype)
    gamma, beta = torch.randn(Cout, generator=g).to(DEV), torch.randn(Cout, generator=g).to(DEV)
    code = ops._contract_code(x, w, Cin)
    fus = lib().gmd_conv3x3_gn_fusable(code, B, H, W, Cin, Cout, 1, 0, 0, 32, ops.WORKSPACE_BYTES)
    cpg_bytes = Cout // 32 * x.element_size()
    # taken when the launch runs as K slices (every case did under the round-3 plans; since round 4 conv 16x16 640->1280 at batch 8
    # has one 64 x 160 loader/consumer tile per CU and is not split), the group slice fits the register-resident kernel and
    # B x G fills the chip
    split_k = dtype == torch.float32 or ops.gemm_plan_info(dtype, B * H * W, Cout, 9 * Cin)[3] > 1
    want = split_k and cpg_bytes % 16 == 0 and H * W * (cpg_bytes // 16) <= 256 * 12 and B * 32 >= 256
    assert bool(fus) == want
    for kw, silu in ((dict(bias=bias, rowbias=(temb, Cout)), True), (dict(bias=bias, residual=res), False), (dict(), True),
                     (dict(rowbias=temb[:, :Cout].contiguous(), residual=res), True)):
        y, _, _ = ops.conv3x3(x, w, B, H, W, **kw)
        yn = ops.groupnorm(y, B, 32, gamma, beta, 1e-5, silu=silu)
        r1, n1 = ops.conv3x3_groupnorm(x, w, B, H, W, 32, gamma, beta, 1e-5, silu=silu, want_raw=True, **kw)
        r0, n0 = ops.conv3x3_groupnorm(x, w, B, H, W, 32, gamma, beta, 1e-5, silu=silu, **kw)
        assert r0 is None and torch.equal(r1, y) and torch.equal(n1, yn) and torch.equal(n0, yn)
        if split_k and cpg_bytes % 16 == 0 and H * W * (cpg_bytes // 16) <= 256 * 12:  # the entry point itself also runs small batches
            r2 = torch.empty_like(y)
            n2 = torch.empty_like(y)
            rb = kw.get("rowbias")
            rb_ptr, rb_ld = ops._rowbias(rb)
            ops.check(lib().gmd_conv3x3_groupnorm(x.data_ptr(), w.data_ptr(), r2.data_ptr(), n2.data_ptr(), code, B, H, W, Cin, Cout, 1, 0, 0,
                                                  ops._ptr(kw.get("bias")), rb_ptr, rb_ld, ops._ptr(kw.get("residual")), float(getattr(w, "_alpha", 1.0)),
                                                  32, 1e-5, gamma.data_ptr(), beta.data_ptr(), int(silu), ops._workspace(x.device).data_ptr(),
                                                  ops.WORKSPACE_BYTES, torch.cuda.current_stream().cuda_stream), "gmd_conv3x3_groupnorm")
            assert torch.equal(r2, y) and torch.equal(n2, yn)
        assert torch.isfinite(yn.float()).all()


def test_conv3x3_groupnorm_refuses_unfusable_launch():
    """Called directly on a launch whose plan is not split-K (64x64 level) the entry point fails loudly instead of running."""
    from gm_diffusion import hip_ops as ops
    from gm_diffusion._native import lib
    B, H, C = 2, 64, 320
    x = torch.randn(B, H * H, C, device=DEV).bfloat16()
    w = (torch.randn(C, 9 * C, device=DEV) * 0.02).bfloat16()
    gamma = torch.ones(C, device=DEV)
    yn = torch.empty_like(x)
    assert lib().gmd_conv3x3_gn_fusable(ops.GMD_BF16, B, H, H, C, C, 1, 0, 0, 32, ops.WORKSPACE_BYTES) == 0
    rc = lib().gmd_conv3x3_groupnorm(x.data_ptr(), w.data_ptr(), None, yn.data_ptr(), ops.GMD_BF16, B, H, H, C, C, 1, 0, 0, None, None, 0, None,
                                     1.0, 32, 1e-5, gamma.data_ptr(), gamma.data_ptr(), 1, ops._workspace(x.device).data_ptr(),
                                     ops.WORKSPACE_BYTES, torch.cuda.current_stream().cuda_stream)
    assert rc == 3 and b"does not fuse" in lib().gmd_last_error()
    torch.cuda.synchronize()
