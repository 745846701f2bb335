"""float16 on the matrix-core path (GMD_F16): the element type the reference's own half-precision runs use
(scripts/stage2/experiments/batch_size_sweep.py: ``.to(device, dtype=torch.float16)``).  Same kernels, layouts and launch
plans as bfloat16; 11 significand bits instead of 8, so every comparison here is ~8x tighter than its bfloat16 twin and the
latent drift over a PNDM trajectory drops by the same factor."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
H16 = torch.float16


def ops():
    from gm_diffusion import hip_ops

    return hip_ops


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_err(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max())


def _attn_ref(q, k, v, heads, scale):
    B, Nq, C = q.shape
    d = C // heads
    qh = q.double().view(B, Nq, heads, d).transpose(1, 2)
    kh = k.double().view(B, -1, heads, d).transpose(1, 2)
    vh = v.double().view(B, -1, heads, d).transpose(1, 2)
    p = torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1)
    return (p @ vh).transpose(1, 2).reshape(B, Nq, C)


@pytest.mark.parametrize("D", [32, 40, 64, 80, 160])
@pytest.mark.parametrize("Nq,Nk", [(256, 256), (64, 77), (200, 130), (1024, 1024)])
def test_attention_f16(D, Nq, Nk):
    o = ops()
    heads, B = 2, 2
    C = heads * D
    g = torch.Generator().manual_seed(D * 7 + Nq + Nk)
    q, k, v = (torch.randn(B, n, C, generator=g).half() for n in (Nq, Nk, Nk))
    ld = (Nk + 7) // 8 * 8
    vt = torch.full((B, C, ld), float("nan")).half()  # pad columns poisoned: the kernel must mask them
    vt[:, :, :Nk] = v.transpose(1, 2)
    got = o.attention(q.to(DEV), k.to(DEV), vt.to(DEV), heads, Nk, D ** -0.5)
    ref = _attn_ref(q, k, v, heads, D ** -0.5)
    assert got.dtype == H16 and torch.isfinite(got.float()).all()
    assert rel_err(got.float(), ref) < 2e-3 and max_err(got.float(), ref) < 1e-2


@pytest.mark.parametrize("spike", [1.5, 3.0, 8.0])
def test_attention_f16_spiked_keys_stay_finite(spike):
    """float16 tops out at 65504: P = 2^(score - lagged stabiliser) must never be converted beyond that.  A key whose score
    jumps far above the running maximum in a later tile takes the rescale (small jump) or the classic-softmax redo of the
    block (jump beyond the 2^14 window) -- never an inf / NaN."""
    o = ops()
    heads, B, D, N = 1, 1, 40, 320
    g = torch.Generator().manual_seed(77)
    q = torch.randn(B, N, D, generator=g)
    k = torch.randn(B, N, D, generator=g) * 0.3
    v = torch.randn(B, N, D, generator=g)
    k[0, 200] = q[0, 5] * spike
    k[0, 310] = q[0, 100] * (spike + 1.0)
    q, k, v = q.half(), k.half(), v.half()
    got = o.attention(q.to(DEV), k.to(DEV), v.transpose(1, 2).contiguous().to(DEV), heads, N, D ** -0.5)
    ref = _attn_ref(q, k, v, heads, D ** -0.5)
    assert torch.isfinite(got.float()).all()
    assert max_err(got.float(), ref) < 1e-2 and rel_err(got.float(), ref) < 2e-3


def test_geglu_epilogue_and_splitk_f16():
    o = ops()
    g = torch.Generator().manual_seed(4)
    M, C = 1024, 320
    x = torch.randn(M, C, generator=g).half()
    w = (torch.randn(8 * C, C, generator=g) / math.sqrt(C)).half()
    b = torch.randn(8 * C, generator=g)
    half = 4 * C  # host re-layout of the fused GEGLU epilogue: value / gate rows interleaved in groups of 16
    wi = torch.stack([w[:half].reshape(half // 16, 16, -1), w[half:].reshape(half // 16, 16, -1)], 1).reshape(2 * half, -1)
    bi = torch.stack([b[:half].reshape(half // 16, 16), b[half:].reshape(half // 16, 16)], 1).reshape(2 * half)
    got = o.gemm_nt(x.to(DEV), wi.contiguous().to(DEV), bias=bi.contiguous().to(DEV), act=o.ACT_GEGLU)
    y = x.double() @ w.double().t() + b.double()
    ref = y[:, :half] * F.gelu(y[:, half:])
    assert got.shape == (M, half) and rel_err(got.float(), ref) < 2e-3
    # deep K, few tiles: the split-K slabs + fixed-order reduction
    a = torch.randn(512, 5120, generator=g).half()
    w2 = (torch.randn(1280, 5120, generator=g) / math.sqrt(5120)).half()
    r = torch.randn(512, 1280, generator=g).half()
    got = o.gemm_nt(a.to(DEV), w2.to(DEV), residual=r.to(DEV))
    assert rel_err(got.float(), a.double() @ w2.double().t() + r.double()) < 2e-3
    # conv3x3 at an 8x8 level (split-K) and at 64x64
    for B, H, ci, co in [(8, 8, 1280, 1280), (2, 64, 320, 320)]:
        xx = torch.randn(B, ci, H, H, generator=g).half()
        wt = (torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci)).half()
        bb = torch.randn(co, generator=g)
        ref = F.conv2d(xx.double(), wt.double(), bb.double(), padding=1).permute(0, 2, 3, 1).reshape(B, H * H, co)
        got, _, _ = o.conv3x3(xx.permute(0, 2, 3, 1).reshape(B, H * H, ci).contiguous().to(DEV),
                              wt.permute(0, 2, 3, 1).reshape(co, 9 * ci).contiguous().to(DEV), B, H, H, bias=bb.to(DEV))
        assert rel_err(got.float(), ref) < 2e-3, (B, H, ci, co)


def _hip(model_cls, oracle_model, dtype):
    m = model_cls(**vars(oracle_model.config))
    m.load_state_dict(oracle_model.state_dict())
    return m.to(DEV, dtype)


def test_unet_and_vae_f16_vs_oracle():
    from gm_diffusion.components import AutoencoderKL, UNet2DConditionModel
    from oracle import fixtures

    g = torch.Generator().manual_seed(3)
    for kind, hw, tol_h, tol_b in (("tiny", 16, 4e-3, 3e-2), ("sd15", 8, 6e-3, 4e-2)):
        ou = fixtures.build_unet(kind, 8)
        x = torch.randn(2, 8, hw, hw, generator=g)
        ctx = torch.randn(2, 77, ou.config.cross_attention_dim, generator=g)
        ref = ou(x, torch.tensor(701), encoder_hidden_states=ctx)[0]
        e = {}
        for dt in (H16, torch.bfloat16):
            got = _hip(UNet2DConditionModel, ou, dt)(x.to(DEV), 701, encoder_hidden_states=ctx.to(DEV), return_dict=False)[0]
            assert got.dtype == torch.float32
            e[dt] = rel_err(got, ref)
        print(f"UNet {kind}: eps rel err float16 {e[H16]:.2e}  bfloat16 {e[torch.bfloat16]:.2e}")
        assert e[H16] < tol_h and e[torch.bfloat16] < tol_b and e[H16] < 0.4 * e[torch.bfloat16]
    ov = fixtures.build_vae("tiny")
    z = torch.randn(2, 4, 8, 8, generator=g) * 3
    got = _hip(AutoencoderKL, ov, H16).decode(z.to(DEV), return_dict=False)[0]
    assert rel_err(got, ov.decode(z)[0]) < 4e-3


def test_dual_pipeline_f16_drift_is_a_fraction_of_bf16(golden_dir):
    """The tiny dual-UNet golden (10 PNDM steps, float32 oracle): float16 latents drift ~8x less than bfloat16 ones; graph
    replay, eager launches and the two-stream overlap stay bit-identical in float16 too."""
    from gm_diffusion.components import AutoencoderKL, PNDMScheduler, UNet2DConditionModel
    from gm_diffusion.pipelines import StableDiffusionDualUNetPipeline
    from oracle import fixtures

    gd = np.load(os.path.join(golden_dir, "pipeline_oracle_dual_tiny.npz"))

    def pipe(dt):
        p = StableDiffusionDualUNetPipeline(
            vae=_hip(AutoencoderKL, fixtures.build_vae("tiny"), dt), text_encoder=None, tokenizer=None,
            unet=_hip(UNet2DConditionModel, fixtures.build_unet("tiny", 4), dt),
            gm_unet=_hip(UNet2DConditionModel, fixtures.build_unet("tiny", 8), dt),
            scheduler=PNDMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", skip_prk_steps=True,
                                    steps_offset=1, set_alpha_to_one=False),
            safety_checker=None, feature_extractor=None, requires_safety_checker=False)
        p.set_progress_bar_config(disable=True)
        return p

    kw = dict(prompt_embeds=torch.from_numpy(gd["prompt_embeds"]).to(DEV), negative_prompt_embeds=torch.from_numpy(gd["negative_prompt_embeds"]).to(DEV),
              latents=torch.from_numpy(gd["latents"]).to(DEV), height=128, width=128, num_inference_steps=10, guidance_scale=7.5,
              output_type="latent")
    rms = lambda a, b: float(((a.double().cpu() - torch.as_tensor(b).double()) ** 2).mean().sqrt())
    ph = pipe(H16)
    ph.co_run_plans = True  # same launch plans with and without the stream overlap: the eager run below is compared bit for bit
    sdr_h, gm_h = ph(**kw)
    sdr_b, gm_b = pipe(torch.bfloat16)(**kw)
    dh, db = (rms(sdr_h, gd["sdr_out"]), rms(gm_h, gd["gm_out"])), (rms(sdr_b, gd["sdr_out"]), rms(gm_b, gd["gm_out"]))
    print(f"latent RMS drift vs fp32 oracle: float16 sdr={dh[0]:.2e} gm={dh[1]:.2e}   bfloat16 sdr={db[0]:.2e} gm={db[1]:.2e}")
    assert dh[0] < 0.35 * db[0] and dh[1] < 0.35 * db[1] and dh[0] < 3e-2 and dh[1] < 3e-2
    ph.use_hip_graphs, ph.overlap_streams = False, False
    e = ph(**kw)
    assert torch.equal(e[0], sdr_h) and torch.equal(e[1], gm_h)
    from gm_diffusion import hdr

    out = hdr.decode_to_hdr(ph.vae, sdr_h, gm_h, qmax=99.0)
    assert torch.isfinite(out["hdr"]).all() and out["hdr_u16"].dtype == torch.uint16
