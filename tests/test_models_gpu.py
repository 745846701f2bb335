"""GPU parity of the HIP-backed UNet / VAE against the CPU oracle (same state-dict, same inputs)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _hip_unet(oracle_unet, dtype):
    from gm_diffusion.components import UNet2DConditionModel

    cfg = {k: v for k, v in vars(oracle_unet.config).items()}
    m = UNet2DConditionModel(**cfg)
    m.load_state_dict(oracle_unet.state_dict())
    return m.to(DEV, dtype)


def _hip_vae(oracle_vae, dtype):
    from gm_diffusion.components import AutoencoderKL

    m = AutoencoderKL(**{k: v for k, v in vars(oracle_vae.config).items()})
    m.load_state_dict(oracle_vae.state_dict())
    return m.to(DEV, dtype)


@pytest.mark.parametrize("in_ch", [4, 8])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
def test_tiny_unet_forward(in_ch, dtype, tol):
    from oracle import fixtures

    ou = fixtures.build_unet("tiny", in_ch)
    hu = _hip_unet(ou, dtype)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, in_ch, 16, 16, generator=g)
    ctx = torch.randn(2, 77, ou.config.cross_attention_dim, generator=g)
    for t in (981, 41):
        ref = ou(x, torch.tensor(t), encoder_hidden_states=ctx)[0]
        got = hu(x.to(DEV), t, encoder_hidden_states=ctx.to(DEV), return_dict=False)[0]
        assert got.dtype == torch.float32 and got.shape == ref.shape
        assert rel_err(got, ref) < tol, f"t={t}"


def test_tiny_unet_non_square_and_tuple_input():
    from oracle import fixtures

    ou = fixtures.build_unet("tiny", 8)
    hu = _hip_unet(ou, torch.float32)
    g = torch.Generator().manual_seed(4)
    a, b = torch.randn(1, 4, 8, 24, generator=g), torch.randn(1, 4, 8, 24, generator=g)
    ctx = torch.randn(1, 77, ou.config.cross_attention_dim, generator=g)
    ref = ou(torch.cat([a, b], 1), torch.tensor(500), encoder_hidden_states=ctx)[0]
    got = hu((a.to(DEV), b.to(DEV)), 500, encoder_hidden_states=ctx.to(DEV), return_dict=False)[0]
    assert rel_err(got, ref) < 2e-5


def test_sd15_unet_forward_f32_small_latent():
    """Full SD-1.5 channel configuration (320/640/1280, d=40/80/160, 8 heads) on an 8x8 latent."""
    from oracle import fixtures

    ou = fixtures.build_unet("sd15", 8)
    hu = _hip_unet(ou, torch.float32)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 8, 8, 8, generator=g)
    ctx = torch.randn(1, 77, 768, generator=g)
    ref = ou(x, torch.tensor(701), encoder_hidden_states=ctx)[0]
    got = hu(x.to(DEV), 701, encoder_hidden_states=ctx.to(DEV), return_dict=False)[0]
    assert rel_err(got, ref) < 3e-5
    hb = _hip_unet(ou, torch.bfloat16)
    got_b = hb(x.to(DEV), 701, encoder_hidden_states=ctx.to(DEV), return_dict=False)[0]
    assert rel_err(got_b, ref) < 4e-2


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
def test_tiny_vae_decode(dtype, tol):
    from oracle import fixtures

    ov = fixtures.build_vae("tiny")
    hv = _hip_vae(ov, dtype)
    z = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(9)) * 3
    ref = ov.decode(z)[0]
    got = hv.decode(z.to(DEV), return_dict=False)[0]
    assert got.shape == ref.shape and rel_err(got, ref) < tol
    nhwc, H, W = hv.decode_nhwc(z.to(DEV))
    assert (H, W) == (64, 64) and nhwc.shape == (2, 64 * 64, 4)
    assert torch.equal(nhwc[..., :3].permute(0, 2, 1).reshape(2, 3, 64, 64), got)


def test_tiny_vae_encode_f32():
    from oracle import fixtures

    ov = fixtures.build_vae("tiny", with_encoder=True)
    hv = _hip_vae(ov, torch.float32)
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(2)) * 2 - 1
    ref = ov.encode(x).latent_dist
    got = hv.encode(x.to(DEV)).latent_dist
    assert rel_err(got.mean, ref.mean) < 2e-5 and rel_err(got.std, ref.std) < 2e-5
    s1 = got.sample(torch.Generator().manual_seed(5)).cpu()
    s2 = ref.sample(torch.Generator().manual_seed(5))
    assert rel_err(s1, s2) < 2e-5


def test_models_refuse_cpu():
    from gm_diffusion._native import HipExtensionError
    from gm_diffusion.components import UNet2DConditionModel
    from oracle import fixtures

    ou = fixtures.build_unet("tiny", 4)
    m = UNet2DConditionModel(**vars(ou.config))
    m.load_state_dict(ou.state_dict())
    with pytest.raises(HipExtensionError):
        m(torch.zeros(1, 4, 8, 8), 1, encoder_hidden_states=torch.zeros(1, 77, 64))


@pytest.mark.parametrize("which,dtype,tol", [("tiny", torch.float32, 3e-5), ("tiny", torch.bfloat16, 4e-2),
                                             ("clip_l", torch.float32, 5e-5), ("clip_l", torch.bfloat16, 4e-2)])
def test_clip_text_encoder_matches_oracle(which, dtype, tol):
    """the HIP CLIP text encoder against the (transformers-pinned) oracle: last hidden state, pooled
    output, every hidden state, and the clip_skip branch's final_layer_norm."""
    from gm_diffusion.components import CLIPTextModel
    from oracle import clip_text as C

    cfg = C.tiny_clip_config() if which == "tiny" else C.clip_l_config()
    torch.manual_seed(11)
    ref = C.CLIPTextModel(**cfg).eval()
    with torch.no_grad():
        for n, p_ in ref.named_parameters():
            if "layer_norm" not in n:
                p_.mul_(1.5)  # livelier logits than the default init
    m = CLIPTextModel(**cfg)
    m.load_state_dict(ref.state_dict())
    m.to(DEV, dtype)
    g = torch.Generator().manual_seed(12)
    ids = torch.randint(3, cfg["vocab_size"] - 1, (3, 77), generator=g)
    ids[0, 9] = ids[1, 76] = ids[2, 40] = cfg["vocab_size"] - 1
    want = ref(ids, output_hidden_states=True)
    got = m(ids.to(DEV), output_hidden_states=True)
    assert got[0].shape == (3, 77, cfg["hidden_size"]) and got[0].dtype == dtype
    assert rel_err(got[0].float(), want[0]) < tol
    assert rel_err(got.pooler_output.float(), want[1]) < tol
    assert len(got.hidden_states) == cfg["num_hidden_layers"] + 1
    for a, b in zip(got[-1], want[2]):
        assert rel_err(a.float(), b) < tol
    skip = m.text_model.final_layer_norm(got[-1][-2])
    assert rel_err(skip.float(), ref.text_model.final_layer_norm(want[2][-2]).detach()) < tol
    with pytest.raises(IndexError):
        m(torch.full((1, 77), cfg["vocab_size"], device=DEV))
    with pytest.raises(Exception):
        m(ids)  # host tensor: no CPU path


def test_vae_decode_sliced_batch_equals_whole():
    """Decoder batches whose widest activation would pass 4 GiB (32-bit buffer offsets) are decoded in slices: same result."""
    from oracle import fixtures

    m = _hip_vae(fixtures.build_vae("tiny"), torch.float32)
    z = torch.randn(3, 4, 8, 8, generator=torch.Generator().manual_seed(2)).to(DEV)
    whole, H, W = m.decode_nhwc(z)
    m.max_tensor_bytes = 64 * 64 * 64 * 4 + 1  # room for one image's widest full-resolution activation only
    parts, H2, W2 = m.decode_nhwc(z)
    assert (H, W) == (H2, W2) == (64, 64) and parts.shape == whole.shape
    assert rel_err(parts, whole) < 1e-5


def test_sd15_vae_decode_small_latent():
    """SD-1.5-width AutoencoderKL decoder (512/512/256/128 channels, the 512-channel single-head mid-block attention) on an
    8x8 latent against the oracle: float32 <= 3e-5, bf16 <= 4e-2 (the full-width VAE was never compared)."""
    from oracle import fixtures

    ov = fixtures.build_vae("sd15")
    z = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(19)) * 3
    ref = ov.decode(z)[0]
    assert ref.shape == (2, 3, 64, 64)
    got = _hip_vae(ov, torch.float32).decode(z.to(DEV), return_dict=False)[0]
    assert got.shape == ref.shape and rel_err(got, ref) < 3e-5
    got_b = _hip_vae(ov, torch.bfloat16).decode(z.to(DEV), return_dict=False)[0]
    assert rel_err(got_b.float(), ref) < 4e-2


def test_sd15_unet_forward_f32_32x32_latent_cfg_batch():
    """Full-width SD-1.5 UNet at the BASELINE config-1 latent size (32x32 = 1024 / 256 / 64 / 16 tokens per level), batch 2
    (the CFG pair), two timesteps: float32 vs the oracle."""
    from oracle import fixtures

    ou = fixtures.build_unet("sd15", 8)
    hu = _hip_unet(ou, torch.float32)
    g = torch.Generator().manual_seed(16)
    x = torch.randn(2, 8, 32, 32, generator=g)
    ctx = torch.randn(2, 77, 768, generator=g)
    for t in (901, 101):
        ref = ou(x, torch.tensor(t), encoder_hidden_states=ctx)[0]
        got = hu(x.to(DEV), t, encoder_hidden_states=ctx.to(DEV), return_dict=False)[0]
        assert rel_err(got, ref) < 3e-5, t


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
def test_unet_cfg_shared_prefix_equals_duplicated_batch(dtype, tol):
    """Classifier-free guidance evaluates the same latents under two text conditionings (stable_diffusion_dual_unet.py:1045-
    1047): computing the layers in front of the first cross-attention once (``cfg_shared``) must give what the duplicated
    batch gives, for both halves, and against the oracle on the duplicated batch."""
    from oracle import fixtures

    ou = fixtures.build_unet("tiny", 4)
    hu = _hip_unet(ou, dtype)
    g = torch.Generator().manual_seed(23)
    lat = torch.randn(3, 4, 16, 16, generator=g)
    ctx = torch.randn(6, 77, ou.config.cross_attention_dim, generator=g)  # [uncond x3, cond x3]
    ref = ou(torch.cat([lat, lat]), torch.tensor(601), encoder_hidden_states=ctx)[0]
    hu._ensure()
    c = hu.prepare_context(ctx.to(DEV))
    hu.set_timestep(601)
    full = hu.forward_packed(hu.pack_input(lat.to(DEV), dup=2), 6, 16, 16, c)
    shared = hu.forward_packed(hu.pack_input(lat.to(DEV), dup=1), 6, 16, 16, c, cfg_shared=True)
    assert shared.shape == full.shape == (6, 4, 16, 16)
    assert rel_err(shared, full) < tol and rel_err(shared, ref) < max(tol, 3e-5 if dtype == torch.float32 else 3e-2)
    assert not torch.equal(shared[:3], shared[3:])  # the halves do differ (different conditioning)
    # captured-graph form: same numbers as the eager shared path, bit for bit
    gph = hu.graphed_forward(6, 16, 16, c, cfg_shared=True)
    hu.pack_input(lat.to(DEV), dup=1, out=gph.x)
    assert torch.equal(gph.replay(), shared)


def test_sd15_unet_bf16_producer_statistics_path_vs_oracle_and_vs_statistics_launch():
    """Full-width SD-1.5 UNet at the benchmark's latent size (64x64, batch 4): the GroupNorms of the two upper levels take
    their statistics from the epilogue of the convolution / projection that produced their input.  Same forward with the
    switch off (separate statistics launches) and the CPU oracle as references."""
    from gm_diffusion import hip_ops as ops
    from oracle import fixtures

    ou = fixtures.build_unet("sd15", 8)
    hu = _hip_unet(ou, torch.bfloat16)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(4, 8, 64, 64, generator=g)
    ctx = torch.randn(4, 77, 768, generator=g)
    ref = ou(x, torch.tensor(501), encoder_hidden_states=ctx)[0]
    before = ops.colstats_uses
    got = hu(x.to(DEV), 501, encoder_hidden_states=ctx.to(DEV), return_dict=False)[0]
    used = ops.colstats_uses - before
    # the GroupNorms of the 64x64 level (at batch 4 the 32x32 level's launches are too small to emit); one of the 16 lost its
    # producer statistics in round 4: the 960->320 convolution at batch 4 runs as two K slices of the ping-pong kernel
    # (126 -> 105 us, tools/sweep_pp.py) and split-K launches emit none
    assert used >= 15, used
    ops.USE_COLSTATS = False
    try:
        plain = hu(x.to(DEV), 501, encoder_hidden_states=ctx.to(DEV), return_dict=False)[0]
        assert ops.colstats_uses - before == used
    finally:
        ops.USE_COLSTATS = True
    e_ref, e_plain = rel_err(got, ref), rel_err(plain, ref)
    assert e_ref < 3e-2 and e_plain < 3e-2 and e_ref < 1.25 * e_plain + 1e-3, (e_ref, e_plain)
    assert rel_err(got, plain) < 1.5e-2


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2.5e-2), (torch.float16, 4e-3)])
def test_cfg_shared_prefix_carries_producer_statistics_through_the_duplication(dtype, tol):
    """Full-width UNet at 64x64 latents, half precision: the CFG shared prefix duplicates tensors that carry their producer's
    GroupNorm statistics (``_dup_batch``); the statistics must be duplicated alike.  Shared vs duplicated batch, with the
    producer statistics on and off, and the statistics path must really be taken."""
    from gm_diffusion import hip_ops as ops
    from oracle import fixtures

    hu = _hip_unet(fixtures.build_unet("sd15", 4), dtype)
    g = torch.Generator().manual_seed(41)
    lat = torch.randn(4, 4, 64, 64, generator=g)  # 4 unique samples: the prefix launches (M = 16384) still fill the chip with 128-row tiles
    ctx = torch.randn(8, 77, 768, generator=g)  # [uncond x4, cond x4]
    hu._ensure()
    c = hu.prepare_context(ctx.to(DEV))
    hu.set_timestep(333)
    out = {}
    for use in (True, False):
        ops.USE_COLSTATS = use
        try:
            before = ops.colstats_uses
            full = hu.forward_packed(hu.pack_input(lat.to(DEV), dup=2), 8, 64, 64, c)
            used_full = ops.colstats_uses - before
            shared = hu.forward_packed(hu.pack_input(lat.to(DEV), dup=1), 8, 64, 64, c, cfg_shared=True)
            used_shared = ops.colstats_uses - before - used_full
        finally:
            ops.USE_COLSTATS = True
        if use:
            # the same GroupNorms are served from producer statistics in both forms: the prefix tensors' statistics survive
            # the duplication (skip connections and the residual stream included)
            assert used_full >= 16 and used_shared == used_full, (used_full, used_shared)
        else:
            assert used_full == 0 and used_shared == 0
        assert rel_err(shared, full) < tol, (use, rel_err(shared, full))
        assert not torch.equal(shared[:4], shared[4:])
        out[use] = shared
    assert rel_err(out[True], out[False]) < tol


# ---------------------------------------------------------------------------------------------
# SDXL-style UNet options (BASELINE.json configs[4]; no reference behaviour exists: parity is against the oracle's restatement)
# ---------------------------------------------------------------------------------------------
def _sdxl_inputs(ou, B, hw, L=9, seed=50):
    c = ou.config
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, c.in_channels, hw, hw, generator=g)
    ctx = torch.randn(B, L, c.cross_attention_dim, generator=g)
    P = c.projection_class_embeddings_input_dim - 6 * c.addition_time_embed_dim
    kw = dict(text_embeds=torch.randn(B, P, generator=g), time_ids=torch.tensor([[1024.0, 768, 0, 16, 1024, 768], [512, 512, 32, 0, 640, 512]][:B] * (B // 2 + 1))[:B])
    return x, ctx, kw


@pytest.mark.parametrize("dtype,mode,tol", [(torch.float32, "split", 3e-5), (torch.float32, "exact", 3e-5), (torch.bfloat16, "split", 4e-2),
                                            (torch.float16, "split", 6e-3)])
def test_tiny_sdxl_style_unet_forward(dtype, mode, tol):
    """Three levels, no attention at the first, transformer depths 1 / 2 / 3 with heads of 64, nn.Linear projections, text_time
    added conditioning -- HIP UNet against the oracle, eager and as a captured graph, with a second set of conditioning values
    replayed through the same graph."""
    from gm_diffusion import hip_ops as ops
    from oracle import fixtures, unet as OU

    prev = ops.set_f32_mode(mode)
    try:
        torch.manual_seed(77)
        ou = OU.UNet2DConditionModel(**OU.tiny_sdxl_unet_config()).eval().requires_grad_(False)
        hu = _hip_unet(ou, dtype)
        x, ctx, kw = _sdxl_inputs(ou, 2, 16)
        ref = ou(x, torch.tensor(333), encoder_hidden_states=ctx, added_cond_kwargs=kw)[0]
        dkw = {k: v.to(DEV) for k, v in kw.items()}
        got = hu(x.to(DEV), 333, encoder_hidden_states=ctx.to(DEV), added_cond_kwargs=dkw, return_dict=False)[0]
        assert got.shape == ref.shape and rel_err(got, ref) < tol, rel_err(got, ref)
        # captured graph + in-place update of the conditioning
        c = hu.prepare_context(ctx.to(DEV))
        hu.set_timestep(333)
        hu.set_added_cond(dkw, 2)
        gph = hu.graphed_forward(2, 16, 16, c)
        hu.pack_input(x.to(DEV), out=gph.x)
        assert torch.equal(gph.replay(), got)
        kw2 = dict(text_embeds=kw["text_embeds"] * 0.5, time_ids=kw["time_ids"] + 64.0)
        ref2 = ou(x, torch.tensor(333), encoder_hidden_states=ctx, added_cond_kwargs=kw2)[0]
        hu.set_added_cond({k: v.to(DEV) for k, v in kw2.items()}, 2)
        got2 = gph.replay()
        assert rel_err(got2, ref2) < tol and rel_err(got2, ref) > 0.05  # the new conditioning was really picked up by the replay
        with pytest.raises(ValueError):
            hu(x.to(DEV), 333, encoder_hidden_states=ctx.to(DEV), return_dict=False)  # the conditioning is mandatory for this UNet
    finally:
        ops.set_f32_mode(prev)


def test_sdxl_unet_full_width_small_latent_vs_oracle():
    """The SDXL-base UNet configuration at full width (2.57 G parameters: widths 320 / 640 / 1280, transformer depths 1 / 2 / 10,
    20 heads of 64, 2048-wide text conditioning) on a 16x16 latent, float32 on the matrix cores and bfloat16, against the oracle."""
    from oracle import unet as OU

    torch.manual_seed(5)
    torch.set_num_threads(16)
    ou = OU.UNet2DConditionModel(**OU.SDXL_UNET_CONFIG).eval().requires_grad_(False)
    x, ctx, kw = _sdxl_inputs(ou, 2, 16, L=77)
    ref = ou(x, torch.tensor(601), encoder_hidden_states=ctx, added_cond_kwargs=kw)[0]
    dkw = {k: v.to(DEV) for k, v in kw.items()}
    hu = _hip_unet(ou, torch.float32)
    got = hu(x.to(DEV), 601, encoder_hidden_states=ctx.to(DEV), added_cond_kwargs=dkw, return_dict=False)[0]
    assert rel_err(got, ref) < 5e-5, rel_err(got, ref)
    del hu
    torch.cuda.empty_cache()
    hb = _hip_unet(ou, torch.bfloat16)
    got_b = hb(x.to(DEV), 601, encoder_hidden_states=ctx.to(DEV), added_cond_kwargs=dkw, return_dict=False)[0]
    assert rel_err(got_b, ref) < 6e-2, rel_err(got_b, ref)
