"""CPU: known-answer tests that pin the oracle's restatement of the un-vendored diffusers arithmetic
(parity with real diffusers is unpinned -- diffusers is not installed; these pin structure and formulas)."""
import math

import numpy as np
import pytest
import torch

from oracle import schedulers as OS, unet as OU, vae as OV


def test_sd15_unet_structure_matches_public_facts():
    with torch.device("meta"):
        m = OU.UNet2DConditionModel()
    n = sum(p.numel() for p in m.parameters())
    assert n == 859_520_964  # parameter count of the SD-1.5 UNet
    assert len(m.state_dict()) == 686
    assert tuple(m.up_blocks[1].resnets[2].conv1.weight.shape) == (1280, 1920, 3, 3)
    assert tuple(m.up_blocks[3].resnets[0].conv1.weight.shape) == (320, 960, 3, 3)
    with torch.device("meta"):
        m8 = OU.UNet2DConditionModel(in_channels=8)
    assert sum(p.numel() for p in m8.parameters()) - n == 320 * 4 * 9


def test_sdxl_unet_structure_matches_public_facts():
    """BASELINE.json configs[4] names SDXL-base-1.0; the reference has no SDXL path, so the oracle's SDXL-style options are
    pinned by public facts only: 2,567,463,684 parameters / 1680 state-dict entries, three levels, transformer depths
    1 / 2 / 10 with heads of 64, linear projections, the text_time addition embedding (2816 -> 1280)."""
    with torch.device("meta"):
        m = OU.UNet2DConditionModel(**OU.SDXL_UNET_CONFIG)
    assert sum(p.numel() for p in m.parameters()) == 2_567_463_684
    sd = m.state_dict()
    assert len(sd) == 1680
    assert tuple(sd["add_embedding.linear_1.weight"].shape) == (1280, 2816)
    assert tuple(sd["down_blocks.1.attentions.0.proj_in.weight"].shape) == (640, 640)  # nn.Linear, not conv1x1
    assert "down_blocks.0.attentions.0.norm.weight" not in sd and "up_blocks.2.attentions.0.norm.weight" not in sd
    assert len(m.mid_block.attentions[0].transformer_blocks) == 10 and len(m.up_blocks[0].attentions[2].transformer_blocks) == 10
    assert len(m.down_blocks[1].attentions[0].transformer_blocks) == 2 and m.down_blocks[1].attentions[0].transformer_blocks[0].attn1.heads == 10
    assert tuple(sd["mid_block.attentions.0.transformer_blocks.9.attn2.to_k.weight"].shape) == (1280, 2048)
    # the product's key table describes the same network
    from gm_diffusion.components.unet_2d_condition import SDXL_UNET_CONFIG, UNet2DConditionModel

    want = UNet2DConditionModel(**SDXL_UNET_CONFIG).expected_keys()
    assert set(want) == set(sd) and all(tuple(sd[k].shape) == tuple(v) for k, v in want.items())


def test_sdxl_style_unet_added_conditioning_known_answer():
    """text_time: with add_embedding zeroed the conditioning must vanish; time_ids enter through the sinusoid of each scalar."""
    torch.manual_seed(0)
    m = OU.UNet2DConditionModel(**OU.tiny_sdxl_unet_config()).eval().requires_grad_(False)
    x, ctx = torch.randn(2, 4, 16, 16), torch.randn(2, 7, 128)
    kw = dict(text_embeds=torch.randn(2, 80), time_ids=torch.tensor([[128.0, 96, 0, 8, 128, 96]] * 2))
    a = m(x, torch.tensor(400), encoder_hidden_states=ctx, added_cond_kwargs=kw)[0]
    kw2 = dict(kw, time_ids=kw["time_ids"] + 1.0)
    assert (m(x, torch.tensor(400), encoder_hidden_states=ctx, added_cond_kwargs=kw2)[0] - a).abs().max() > 1e-6
    for p in m.add_embedding.parameters():
        p.zero_()
    b = m(x, torch.tensor(400), encoder_hidden_states=ctx, added_cond_kwargs=kw)[0]
    b2 = m(x, torch.tensor(400), encoder_hidden_states=ctx, added_cond_kwargs=kw2)[0]
    assert torch.equal(b, b2)


def test_sd15_vae_structure_matches_public_facts():
    with torch.device("meta"):
        v = OV.AutoencoderKL(with_encoder=True)
    assert sum(p.numel() for p in v.parameters()) == 83_653_863  # SD-1.5 VAE (encoder + decoder)


def test_timestep_embedding_formula():
    t = torch.tensor([0.0, 1.0, 981.0])
    e = OU.timestep_embedding(t, 320)
    assert e.shape == (3, 320)
    assert torch.allclose(e[0, :160], torch.ones(160)) and torch.allclose(e[0, 160:], torch.zeros(160))  # [cos | sin]
    k = 37
    f = math.exp(-math.log(10000) * k / 160)
    assert abs(float(e[2, k]) - math.cos(981 * f)) < 1e-4 and abs(float(e[2, 160 + k]) - math.sin(981 * f)) < 1e-4


def test_pndm_timesteps_and_closed_form():
    s = OS.PNDMScheduler()
    s.set_timesteps(50)
    ts = s.timesteps.tolist()
    assert len(ts) == 51 and ts[:4] == [981, 961, 961, 941] and ts[-1] == 1
    # eps == 0 for every step: x_prev = sqrt(a_prev/a_t) x, so the product telescopes to sqrt(a_final/a_981)
    x = torch.ones(1, 4, 2, 2)
    for t in s.timesteps:
        x = s.step(torch.zeros_like(x), t, x, return_dict=False)[0]
    a = s.alphas_cumprod
    expect = math.sqrt(float(s.final_alpha_cumprod) / float(a[981]))
    assert abs(float(x.flatten()[0]) - expect) < 1e-4 * expect
    # a consistent-noise model (eps = the true noise of x_t = sqrt(a) x0 + sqrt(1-a) n) is integrated exactly by
    # every PLMS order: the sampler must return x0's trajectory endpoint sqrt(a_0) x0 + sqrt(1-a_0) n
    s.set_timesteps(20)
    x0, n = torch.full((1, 4, 2, 2), 0.7), torch.full((1, 4, 2, 2), -1.3)
    t0 = int(s.timesteps[0])
    x = a[t0].sqrt() * x0 + (1 - a[t0]).sqrt() * n
    for t in s.timesteps:
        x = s.step(n.clone(), t, x, return_dict=False)[0]
    fa = s.final_alpha_cumprod
    assert torch.allclose(x, fa.sqrt() * x0 + (1 - fa).sqrt() * n, atol=2e-4)


def test_ddpm_posterior_mean_known_answer():
    s = OS.DDPMScheduler()
    s.set_timesteps(10)
    assert s.timesteps.tolist() == [901, 801, 701, 601, 501, 401, 301, 201, 101, 1]
    g = torch.Generator().manual_seed(0)
    x, e = torch.randn(1, 4, 2, 2, generator=g), torch.randn(1, 4, 2, 2, generator=g)
    out = s.step(e, 1, x, generator=torch.Generator().manual_seed(1), return_dict=False)[0]
    # t=1 -> prev_t=-99 -> alpha_prev = 1: posterior mean is exactly the predicted x0, plus noise of variance ~0
    a = s.alphas_cumprod[1]
    x0 = (x - (1 - a).sqrt() * e) / a.sqrt()
    assert torch.allclose(out, x0, atol=1e-5)


def test_attention_and_blocks_against_torch_primitives():
    torch.manual_seed(0)
    att = OU.Attention(64, 2)
    x = torch.randn(2, 10, 64)
    q, k, v = att.to_q(x), att.to_k(x), att.to_v(x)
    ref = torch.nn.functional.scaled_dot_product_attention(
        q.view(2, 10, 2, 32).transpose(1, 2), k.view(2, 10, 2, 32).transpose(1, 2), v.view(2, 10, 2, 32).transpose(1, 2))
    ref = att.to_out[0](ref.transpose(1, 2).reshape(2, 10, 64))
    assert torch.allclose(att(x), ref, atol=1e-5)
    geglu = OU.GEGLU(16, 32)
    y = geglu.proj(torch.ones(1, 16))
    assert torch.allclose(geglu(torch.ones(1, 16)), y[:, :32] * torch.nn.functional.gelu(y[:, 32:]))


def test_tiny_models_run_and_are_deterministic():
    from oracle import fixtures

    u1, u2 = fixtures.build_unet("tiny", 8), fixtures.build_unet("tiny", 8)
    assert all(torch.equal(a, b) for a, b in zip(u1.state_dict().values(), u2.state_dict().values()))
    x = torch.randn(1, 8, 8, 8)
    y = u1(x, torch.tensor(3), encoder_hidden_states=torch.randn(1, 77, 64))[0]
    assert y.shape == (1, 4, 8, 8) and torch.isfinite(y).all()
    v = fixtures.build_vae("tiny", with_encoder=True)
    img = v.decode(torch.randn(1, 4, 8, 8))[0]
    assert img.shape == (1, 3, 64, 64)
    assert v.encode(img).latent_dist.mean.shape == (1, 4, 8, 8)


def test_pipeline_goldens_reproduce(golden_dir):
    """The committed oracle pipeline vectors are regenerated bit for bit (guards the fixtures against drift)."""
    import os

    from oracle import fixtures

    g = np.load(os.path.join(golden_dir, "pipeline_oracle_dual_tiny_rescale.npz"))
    out = fixtures.fixture_dual_tiny_rescale()
    assert np.array_equal(out["latents"], g["latents"])
    assert np.allclose(out["sdr_out"], g["sdr_out"], atol=1e-5) and np.allclose(out["gm_out"], g["gm_out"], atol=1e-5)


@pytest.mark.parametrize("which", ["tiny", "clip_l"])
def test_clip_text_oracle_matches_transformers(which):
    """the CLIP text encoder restatement is pinned against the installed ``transformers.CLIPTextModel``
    (random weights from a config; the SD-1.5 text tower shape once)."""
    tr = pytest.importorskip("transformers")
    from oracle import clip_text as C

    cfg = C.tiny_clip_config() if which == "tiny" else C.clip_l_config()
    tcfg = tr.CLIPTextConfig(**{k: v for k, v in cfg.items()}, bos_token_id=0, pad_token_id=1)
    torch.manual_seed(3)
    ref = tr.CLIPTextModel(tcfg).eval()
    mine = C.CLIPTextModel(**cfg).eval()
    mine.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(4)
    ids = torch.randint(3, cfg["vocab_size"] - 1, (2, 77), generator=g)
    ids[0, 20] = cfg["vocab_size"] - 1   # "EOS" = largest id (legacy eos rule), padded after it
    ids[1, 76] = cfg["vocab_size"] - 1
    with torch.no_grad():
        r = ref(ids, output_hidden_states=True)
    o = mine(ids, output_hidden_states=True)
    assert torch.allclose(o[0], r.last_hidden_state, atol=2e-5, rtol=1e-5)
    assert torch.allclose(o[1], r.pooler_output, atol=2e-5, rtol=1e-5)
    assert len(o[2]) == len(r.hidden_states) == cfg["num_hidden_layers"] + 1
    for a, b in zip(o[2], r.hidden_states):
        assert torch.allclose(a, b, atol=2e-5, rtol=1e-5)
    if which == "clip_l":
        assert sum(p.numel() for p in mine.parameters()) == 123_060_480  # public figure for the CLIP ViT-L/14 text tower
