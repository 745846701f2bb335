"""CPU: host-side logic of the drop-in pipelines (argument checks, scheduler protocol, batched GM
embedding slice, callbacks, return types) driven with the ORACLE's UNet as a duck-typed model --
the generic protocol path executes the reference's torch expressions on host tensors -- and compared
with the oracle's own loops and the committed golden vectors."""
import copy
import os

import numpy as np
import pytest
import torch

from gm_diffusion.components import DDPMScheduler, FrozenDict, PNDMScheduler, StableDiffusionPipelineOutput, randn_tensor
from gm_diffusion.pipelines import (StableDiffusionDualUNetImprovedPipeline, StableDiffusionDualUNetPipeline,
                                    StableDiffusionGMPipeline, rescale_noise_cfg, retrieve_timesteps)
from oracle import fixtures, pipelines as OP, schedulers as OS

SD_PNDM = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", skip_prk_steps=True, steps_offset=1)


class FakeVae:
    class config:
        block_out_channels = [1, 2, 3, 4]
        scaling_factor = 0.18215


def gm_pipe(unet, sched=None):
    return StableDiffusionGMPipeline(vae=FakeVae(), text_encoder=None, tokenizer=None, unet=unet,
                                     scheduler=sched or PNDMScheduler(**SD_PNDM), safety_checker=None, feature_extractor=None,
                                     requires_safety_checker=False)


def dual_pipe(unet, gm_unet, sched=None, cls=StableDiffusionDualUNetPipeline):
    return cls(vae=FakeVae(), text_encoder=None, tokenizer=None, unet=unet, gm_unet=gm_unet,
               scheduler=sched or PNDMScheduler(**SD_PNDM), safety_checker=None, feature_extractor=None, requires_safety_checker=False)


@pytest.fixture(scope="module")
def unets():
    return fixtures.build_unet("tiny", 4), fixtures.build_unet("tiny", 8)


def test_pndm_product_equals_oracle_bitwise():
    p, o = PNDMScheduler(**SD_PNDM), OS.PNDMScheduler()
    for n in (50, 10, 7):
        p.set_timesteps(n)
        o.set_timesteps(n)
        assert torch.equal(p.timesteps, o.timesteps) and len(p.timesteps) == n + 1
        g = torch.Generator().manual_seed(n)
        x = torch.randn(2, 4, 4, 4, generator=g)
        xo = x.clone()
        for t in o.timesteps:
            e = torch.randn(2, 4, 4, 4, generator=g)
            x = p.step(e, t, x, return_dict=False)[0]
            xo = o.step(e, t, xo, return_dict=False)[0]
            assert torch.equal(x, xo)
    assert p.step(e, 1, x).prev_sample.shape == x.shape  # return_dict form


def test_scheduler_protocol_and_config():
    s = PNDMScheduler(**SD_PNDM)
    assert s.config.steps_offset == 1 and s.config["beta_schedule"] == "scaled_linear" and s.order == 1 and s.init_noise_sigma == 1.0
    assert isinstance(s.config, FrozenDict)
    with pytest.raises(TypeError):
        s.config["steps_offset"] = 2
    s.set_timesteps(5)
    s2 = copy.deepcopy(s)
    assert torch.equal(s2.timesteps, s.timesteps) and s2.config == s.config and s2 is not s
    d = DDPMScheduler.from_config(s.config)  # scheduler swap by from_config (formal_improved.py:195)
    assert d.config.beta_start == 0.00085 and d.config.steps_offset == 1
    with pytest.raises(NotImplementedError):
        PNDMScheduler(skip_prk_steps=False).set_timesteps(10)
    with pytest.raises(ValueError):
        PNDMScheduler(**SD_PNDM).step(torch.zeros(1), 1, torch.zeros(1))


def test_ddpm_product_equals_oracle():
    p, o = DDPMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1, clip_sample=False), OS.DDPMScheduler()
    p.set_timesteps(8)
    o.set_timesteps(8)
    assert torch.equal(p.timesteps, o.timesteps)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, 4, 4, generator=g)
    xo = x.clone()
    gp, go = torch.Generator().manual_seed(9), torch.Generator().manual_seed(9)
    for t in o.timesteps:
        e = torch.randn(1, 4, 4, 4, generator=g)
        x = p.step(e, t, x, generator=gp, return_dict=False)[0]
        xo = o.step(e, t, xo, generator=go, return_dict=False)[0]
        assert torch.allclose(x, xo, atol=1e-6)


def test_constructor_patches_outdated_scheduler_config(unets):
    s = PNDMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", skip_prk_steps=True, steps_offset=0, clip_sample=True)
    with pytest.warns(FutureWarning):
        p = gm_pipe(unets[1], s)
    assert p.scheduler.config.steps_offset == 1 and p.scheduler.config.clip_sample is False
    assert p.vae_scale_factor == 8


def test_retrieve_timesteps_errors():
    s = PNDMScheduler(**SD_PNDM)
    ts, n = retrieve_timesteps(s, 10, "cpu")
    assert n == 10 and len(ts) == 11
    with pytest.raises(ValueError):
        retrieve_timesteps(s, None, "cpu", timesteps=[1, 2], sigmas=[0.1])
    with pytest.raises(ValueError):
        retrieve_timesteps(s, None, "cpu", timesteps=[10, 1])  # PNDM.set_timesteps takes no `timesteps`
    with pytest.raises(ValueError):
        retrieve_timesteps(s, None, "cpu", sigmas=[1.0])


def test_rescale_noise_cfg_matches_oracle():
    g = torch.Generator().manual_seed(0)
    a, b = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 4, 8, 8, generator=g)
    assert torch.equal(rescale_noise_cfg(a, b, 0.7), OP.rescale_noise_cfg(a, b, 0.7))


def test_check_inputs_errors(unets):
    p = gm_pipe(unets[1])
    pe = torch.zeros(1, 77, 64)
    lat = torch.zeros(1, 4, 8, 8)
    with pytest.raises(ValueError, match="divisible by 8"):
        p(lat, prompt_embeds=pe, height=12, width=64)
    with pytest.raises(ValueError, match="Provide either"):
        p(lat)
    with pytest.raises(ValueError, match="Cannot forward both"):
        p(lat, prompt="x", prompt_embeds=pe)
    with pytest.raises(ValueError, match="must have the same shape"):
        p(lat, prompt_embeds=pe, negative_prompt_embeds=torch.zeros(1, 70, 64))
    with pytest.raises(ValueError, match="callback_on_step_end_tensor_inputs"):
        p(lat, prompt_embeds=pe, negative_prompt_embeds=pe, callback_on_step_end_tensor_inputs=["nope"])
    with pytest.raises(ValueError, match="callback_steps"):
        p(lat, prompt_embeds=pe, negative_prompt_embeds=pe, callback_steps=0)
    with pytest.raises(ValueError, match="list of generators"):
        p.prepare_latents(2, 4, 64, 64, torch.float32, "cpu", [torch.Generator()])
    with pytest.raises(ValueError, match="has to be of type"):
        p(lat, prompt=3)


def test_gm_pipeline_generic_path_matches_golden(unets, golden_dir):
    g = np.load(os.path.join(golden_dir, "pipeline_oracle_gm_tiny.npz"))
    p = gm_pipe(unets[1])
    p.set_progress_bar_config(disable=True)
    seen = []
    out = p(torch.from_numpy(g["sdr_latent"]), prompt_embeds=torch.from_numpy(g["prompt_embeds"]),
            negative_prompt_embeds=torch.from_numpy(g["negative_prompt_embeds"]), latents=torch.from_numpy(g["latents"]),
            num_inference_steps=10, guidance_scale=7.5, output_type="latent", noise_level=0.0,  # unknown kwargs are ignored
            callback_on_step_end=lambda pipe, i, t, kw: (seen.append(i) or {}))
    assert isinstance(out, StableDiffusionPipelineOutput) and out.nsfw_content_detected is None
    assert torch.equal(out.images, torch.from_numpy(g["out"]))  # same torch expressions on the same host -> bit identical
    assert seen == list(range(11)) and p.num_timesteps == 11
    assert out[0] is out.images
    tup = p(torch.from_numpy(g["sdr_latent"]), prompt_embeds=torch.from_numpy(g["prompt_embeds"]),
            negative_prompt_embeds=torch.from_numpy(g["negative_prompt_embeds"]), latents=torch.from_numpy(g["latents"]),
            num_inference_steps=10, output_type="latent", return_dict=False)
    assert isinstance(tup, tuple) and torch.equal(tup[0], out.images) and tup[1] is None


def test_gm_pipeline_callback_can_replace_latents_and_interrupt(unets):
    p = gm_pipe(unets[1])
    p.set_progress_bar_config(disable=True)
    pe, ne = torch.zeros(1, 77, 64), torch.zeros(1, 77, 64)

    def cb(pipe, i, t, kw):
        if i == 1:
            pipe._interrupt = True
        return {"latents": torch.full_like(kw["latents"], 2.0)}

    out = p(torch.zeros(1, 4, 8, 8), prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=4, output_type="latent",
            generator=torch.Generator().manual_seed(0), callback_on_step_end=cb).images
    assert torch.equal(out, torch.full_like(out, 2.0)) and p.interrupt


def test_dual_pipeline_generic_path_matches_golden(unets, golden_dir):
    g = np.load(os.path.join(golden_dir, "pipeline_oracle_dual_tiny.npz"))
    for cls in (StableDiffusionDualUNetPipeline, StableDiffusionDualUNetImprovedPipeline):
        p = dual_pipe(*unets, cls=cls)
        p.set_progress_bar_config(disable=True)
        out = p(prompt_embeds=torch.from_numpy(g["prompt_embeds"]), negative_prompt_embeds=torch.from_numpy(g["negative_prompt_embeds"]),
                latents=torch.from_numpy(g["latents"]), height=128, width=128, num_inference_steps=10, guidance_scale=7.5,
                output_type="latent", return_dict=True)  # return_dict is ignored: always the bare tuple (dual.py:1132)
        assert isinstance(out, tuple) and len(out) == 2
        assert torch.equal(out[0], torch.from_numpy(g["sdr_out"])) and torch.equal(out[1], torch.from_numpy(g["gm_out"]))
        assert p.gm_scheduler is not p.scheduler and p.gm_scheduler.counter == p.scheduler.counter == 11


def test_dual_pipeline_rescale_and_no_cfg(unets, golden_dir):
    g = np.load(os.path.join(golden_dir, "pipeline_oracle_dual_tiny_rescale.npz"))
    p = dual_pipe(*unets)
    p.set_progress_bar_config(disable=True)
    a, b = p(prompt_embeds=torch.from_numpy(g["prompt_embeds"]), negative_prompt_embeds=torch.from_numpy(g["negative_prompt_embeds"]),
             latents=torch.from_numpy(g["latents"]), height=128, width=128, num_inference_steps=6, guidance_scale=5.0,
             guidance_rescale=0.7, output_type="latent")
    assert torch.equal(a, torch.from_numpy(g["sdr_out"])) and torch.equal(b, torch.from_numpy(g["gm_out"]))
    # guidance_scale <= 1: no CFG, the GM UNet receives the same (only) embeddings -- works, unlike the reference's [1:] slice
    pe = torch.from_numpy(g["prompt_embeds"])
    a1, b1 = p(prompt_embeds=pe, latents=torch.from_numpy(g["latents"]), height=128, width=128, num_inference_steps=3,
               guidance_scale=1.0, output_type="latent")
    ra, rb = OP.dual_loop(unets[0], unets[1], OS.PNDMScheduler(), pe, None, torch.from_numpy(g["latents"]), 3, guidance_scale=1.0)
    assert torch.equal(a1, ra) and torch.equal(b1, rb)
    assert not p.do_classifier_free_guidance and p.guidance_scale == 1.0


def test_dual_pipeline_shared_generator_order_with_ddpm(unets):
    """Stochastic scheduler: both streams draw from the SAME generator, SDR first then GM (dual.py:1015,1077,1093)."""
    sched = DDPMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1, clip_sample=False)
    p = dual_pipe(*unets, sched=sched)
    p.set_progress_bar_config(disable=True)
    pe, ne, lat = fixtures.make_inputs(1, 8, 8, cross_dim=64)
    a, b = p(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, height=64, width=64, num_inference_steps=4,
             generator=torch.Generator().manual_seed(3), output_type="latent")
    ra, rb = OP.dual_loop(unets[0], unets[1], OS.DDPMScheduler(), pe, ne, lat, 4, generator=torch.Generator().manual_seed(3))
    assert torch.allclose(a, ra, atol=1e-5) and torch.allclose(b, rb, atol=1e-5)


def test_randn_tensor_is_shard_independent():
    g1, g2 = torch.Generator().manual_seed(42), torch.Generator().manual_seed(42)
    full = randn_tensor((8, 4, 4, 4), generator=g1, device="cpu", dtype=torch.float32)
    again = randn_tensor((8, 4, 4, 4), generator=g2, device="cpu", dtype=torch.float32)
    assert torch.equal(full, again)
    lst = randn_tensor((2, 4, 4, 4), generator=[torch.Generator().manual_seed(1), torch.Generator().manual_seed(2)])
    assert torch.equal(lst[0], torch.randn((1, 4, 4, 4), generator=torch.Generator().manual_seed(1))[0])


def test_encode_prompt_with_duck_typed_text_encoder(unets):
    class Tok:
        model_max_length = 77

        def __call__(self, text, padding=None, max_length=None, truncation=None, return_tensors=None):
            text = [text] if isinstance(text, str) else text
            n = max_length or 77
            ids = torch.tensor([[len(t) % 50 + 1] * n for t in text])
            return type("E", (), {"input_ids": ids, "attention_mask": torch.ones_like(ids)})()

        def batch_decode(self, ids):
            return [""]

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.emb = torch.nn.Embedding(64, 64)
            self.config = type("C", (), {})()

        @property
        def dtype(self):
            return torch.float32

        def forward(self, ids, attention_mask=None, output_hidden_states=None):
            return (self.emb(ids),)

    p = StableDiffusionDualUNetPipeline(vae=FakeVae(), text_encoder=Enc(), tokenizer=Tok(), unet=unets[0], gm_unet=unets[1],
                                        scheduler=PNDMScheduler(**SD_PNDM), safety_checker=None, feature_extractor=None,
                                        requires_safety_checker=False)
    pe, ne = p.encode_prompt(["a", "bb"], "cpu", 2, True)
    assert pe.shape == (4, 77, 64) and ne.shape == (4, 77, 64)
    assert torch.equal(pe[0], pe[1]) and not torch.equal(pe[0], pe[2])  # repeat per image, prompt-major
    with pytest.raises(TypeError):
        p.encode_prompt("a", "cpu", 1, True, negative_prompt=["x"])
    with pytest.raises(ValueError):
        p.encode_prompt(["a", "b"], "cpu", 1, True, negative_prompt=["x"])
    p.set_progress_bar_config(disable=True)
    sdr, gm = p(prompt=["a", "bb"], height=64, width=64, num_inference_steps=2, output_type="latent",
                generator=torch.Generator().manual_seed(0))
    assert sdr.shape == (2, 4, 8, 8) and gm.shape == (2, 4, 8, 8)


def test_unet_component_key_surface_matches_oracle():
    from gm_diffusion.components import AutoencoderKL, UNet2DConditionModel

    for inc in (4, 8):
        ou = fixtures.build_unet("tiny", inc)
        hu = UNet2DConditionModel(**vars(ou.config))
        assert {k: tuple(v.shape) for k, v in ou.state_dict().items()} == {k: tuple(v) for k, v in hu.expected_keys().items()}
        hu.load_state_dict(ou.state_dict())
        with pytest.raises(KeyError):
            hu.load_state_dict({"conv_in.weight": torch.zeros(1)})
    ov = fixtures.build_vae("tiny", with_encoder=True)
    hv = AutoencoderKL(**vars(ov.config))
    hv.load_state_dict(ov.state_dict())
    assert {k: tuple(v.shape) for k, v in ov.state_dict().items()} == {k: tuple(v) for k, v in hv.expected_keys().items()}
    # full SD-1.5 key surface (686 keys) without allocating weights
    with torch.device("meta"):
        from oracle import unet as OU

        full = OU.UNet2DConditionModel()
    assert {k: tuple(v.shape) for k, v in full.state_dict().items()} == {k: tuple(v) for k, v in UNet2DConditionModel().expected_keys().items()}


def test_checkpoint_directory_roundtrip(tmp_path):
    """diffusers directory layout: model_index.json + <component>/config.json + safetensors."""
    import json

    from gm_diffusion.components import AutoencoderKL, UNet2DConditionModel

    ou = fixtures.build_unet("tiny", 8)
    hu = UNet2DConditionModel(**vars(ou.config))
    hu.load_state_dict(ou.state_dict())
    hu.save_pretrained(str(tmp_path / "unet"))
    cfg = json.load(open(tmp_path / "unet" / "config.json"))
    cfg["num_attention_heads"] = cfg.pop("attention_head_dim")  # the rename generate_hdr.py:99-113 undoes on disk
    json.dump(cfg, open(tmp_path / "unet" / "config.json", "w"))
    back = UNet2DConditionModel.from_pretrained(str(tmp_path), subfolder="unet", in_channels=8)
    assert back.config.attention_head_dim == 2 and back.config.in_channels == 8
    assert all(torch.equal(back.state_dict()[k], v) for k, v in ou.state_dict().items())
    os.makedirs(tmp_path / "scheduler")
    json.dump({"_class_name": "PNDMScheduler", "beta_start": 0.00085, "beta_end": 0.012, "beta_schedule": "scaled_linear",
               "skip_prk_steps": True, "steps_offset": 1, "num_train_timesteps": 1000, "set_alpha_to_one": False,
               "trained_betas": None, "clip_sample": False}, open(tmp_path / "scheduler" / "scheduler_config.json", "w"))
    s = PNDMScheduler.from_pretrained(str(tmp_path), subfolder="scheduler")
    assert s.config.skip_prk_steps and s.config.steps_offset == 1
    d = DDPMScheduler.from_pretrained(str(tmp_path), subfolder="scheduler")  # generate_hdr.py:162 loads DDPM from the PNDM config
    assert d.config.beta_schedule == "scaled_linear"


def test_replace_conv_in_matches_reference_recipe():
    from gm_diffusion.components import UNet2DConditionModel

    ou = fixtures.build_unet("tiny", 4)
    hu = UNet2DConditionModel(**vars(ou.config))
    hu.load_state_dict(ou.state_dict())
    w4, b4 = ou.state_dict()["conv_in.weight"], ou.state_dict()["conv_in.bias"]
    hu.replace_conv_in(8)
    assert hu.config.in_channels == 8
    assert torch.equal(hu.state_dict()["conv_in.weight"], w4.repeat(1, 2, 1, 1) * 0.5)  # generate_hdr.py:79-81
    assert torch.equal(hu.state_dict()["conv_in.bias"], b4)


def test_dpm_solver_pp_known_answer_and_oracle_agreement():
    """DPM-Solver++ is exact for an x0-consistent model: with eps = (x - alpha x0)/sigma the sampler must land on x0
    (sigma_last = 0); product and oracle restatements agree step by step; from_config swap as formal_improved.py:195."""
    from gm_diffusion.components import DPMSolverMultistepScheduler

    ddpm = DDPMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1, clip_sample=False)
    p = DPMSolverMultistepScheduler.from_config(ddpm.config)
    assert p.config.timestep_spacing == "leading" and p.config.steps_offset == 1 and p.config.beta_schedule == "scaled_linear"
    o = OS.DPMSolverMultistepScheduler()
    for n in (20, 8):
        p.set_timesteps(n)
        o.set_timesteps(n)
        assert torch.equal(p.timesteps, o.timesteps) and torch.equal(p.sigmas, o.sigmas) and len(p.timesteps) == n
        x0 = torch.full((1, 4, 2, 2), 0.6)
        noise = torch.full((1, 4, 2, 2), -0.9)
        a0, s0 = p._sigma_to_alpha_sigma_t(p.sigmas[0])
        x = a0 * x0 + s0 * noise
        xo = x.clone()
        for i, t in enumerate(p.timesteps):
            a, s_ = p._sigma_to_alpha_sigma_t(p.sigmas[i])
            eps = (x - a * x0) / s_
            x = p.step(eps, t, x, return_dict=False)[0]
            xo = o.step((xo - a * x0) / s_, t, xo, return_dict=False)[0]
            assert torch.allclose(x, xo, atol=1e-6)
        assert torch.allclose(x, x0, atol=1e-4)
    import inspect

    assert "generator" in inspect.signature(p.step).parameters and "eta" not in inspect.signature(p.step).parameters
    with pytest.raises(NotImplementedError):
        DPMSolverMultistepScheduler(use_karras_sigmas=True)


def test_dual_pipeline_with_dpm_solver(unets):
    from gm_diffusion.components import DPMSolverMultistepScheduler

    sched = DPMSolverMultistepScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1,
                                        timestep_spacing="leading")
    p = dual_pipe(*unets, sched=sched)
    p.set_progress_bar_config(disable=True)
    pe, ne, lat = fixtures.make_inputs(1, 8, 8, cross_dim=64)
    a, b = p(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, height=64, width=64, num_inference_steps=5,
             guidance_scale=9.0, eta=0.7, output_type="latent")  # eta is dropped: step() takes none (gm.py:610-625)
    ra, rb = OP.dual_loop(unets[0], unets[1], OS.DPMSolverMultistepScheduler(), pe, ne, lat, 5, guidance_scale=9.0)
    assert torch.allclose(a, ra, atol=1e-5) and torch.allclose(b, rb, atol=1e-5)


def test_pipeline_from_pretrained_directory(tmp_path):
    """Whole-pipeline load as scripts/inference/generate_hdr.py:152-176 does it: an SD-1.5-layout directory
    (model_index.json + unet/ vae/ text_encoder/ scheduler/), components passed as keyword arguments replace the ones on disk,
    ``["transformers", "CLIPTextModel"]`` resolves to the HIP text encoder, gm_unet must be given explicitly."""
    import json

    from safetensors.torch import save_file

    from gm_diffusion.components import AutoencoderKL, CLIPTextModel, UNet2DConditionModel
    from oracle import clip_text as C

    root = tmp_path / "ckpt"
    ou, ov = fixtures.build_unet("tiny", 4), fixtures.build_vae("tiny")
    hu = UNet2DConditionModel(**vars(ou.config))
    hu.load_state_dict(ou.state_dict())
    hu.save_pretrained(str(root / "unet"))
    os.makedirs(root / "vae")
    json.dump({"_class_name": "AutoencoderKL", **{k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(ov.config).items()}},
              open(root / "vae" / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in ov.state_dict().items()}, str(root / "vae" / "diffusion_pytorch_model.safetensors"))
    tcfg = C.tiny_clip_config()
    ote = C.CLIPTextModel(**tcfg)
    os.makedirs(root / "text_encoder")
    json.dump({"architectures": ["CLIPTextModel"], **tcfg}, open(root / "text_encoder" / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in ote.state_dict().items()}, str(root / "text_encoder" / "model.safetensors"))
    os.makedirs(root / "scheduler")
    json.dump({"_class_name": "PNDMScheduler", "beta_start": 0.00085, "beta_end": 0.012, "beta_schedule": "scaled_linear",
               "skip_prk_steps": True, "steps_offset": 1, "num_train_timesteps": 1000, "set_alpha_to_one": False},
              open(root / "scheduler" / "scheduler_config.json", "w"))
    json.dump({"_class_name": "StableDiffusionPipeline", "unet": ["diffusers", "UNet2DConditionModel"], "vae": ["diffusers", "AutoencoderKL"],
               "text_encoder": ["transformers", "CLIPTextModel"], "tokenizer": ["transformers", "CLIPTokenizer"],
               "scheduler": ["diffusers", "PNDMScheduler"], "safety_checker": ["stable_diffusion", "StableDiffusionSafetyChecker"],
               "feature_extractor": ["transformers", "CLIPImageProcessor"], "requires_safety_checker": True},
              open(root / "model_index.json", "w"))
    gm = UNet2DConditionModel(**vars(ou.config)).load_state_dict(ou.state_dict()).replace_conv_in(8)
    with pytest.raises(ValueError, match="gm_unet"):
        StableDiffusionDualUNetPipeline.from_pretrained(str(root), tokenizer=None, safety_checker=None)
    pipe = StableDiffusionDualUNetPipeline.from_pretrained(str(root), gm_unet=gm, tokenizer=None, safety_checker=None,
                                                           requires_safety_checker=False)
    assert isinstance(pipe.unet, UNet2DConditionModel) and isinstance(pipe.vae, AutoencoderKL) and isinstance(pipe.text_encoder, CLIPTextModel)
    assert isinstance(pipe.scheduler, PNDMScheduler) and pipe.gm_unet is gm and pipe.safety_checker is None and pipe.tokenizer is None
    assert pipe.gm_unet.config.in_channels == 8 and pipe.unet.config.in_channels == 4
    assert all(torch.equal(pipe.text_encoder.state_dict()["text_model." + k if not k.startswith("text_model.") else k], v)
               for k, v in ote.state_dict().items())
    assert pipe.vae_scale_factor == 8 and pipe.text_encoder.config.hidden_size == tcfg["hidden_size"]
    # single-UNet pipeline from the same directory with the 8-channel UNet passed in (generate_hdr.py:169-176)
    gp = StableDiffusionGMPipeline.from_pretrained(str(root), unet=gm, tokenizer=None, safety_checker=None, requires_safety_checker=False)
    assert gp.unet is gm and isinstance(gp.vae, AutoencoderKL)


def test_bench_self_spawns_ranks_without_touching_the_gpu():
    """`python bench.py --gpus 2` with no RANK in the environment: the parent must start 2 ranks through
    torch.distributed.run BEFORE any GPU call.  Without a GPU each rank stops at bench.py's own
    "needs an MI355X" check -- which proves the ranks were started and probed the device themselves."""
    import subprocess
    import sys

    if torch.cuda.is_available():
        pytest.skip("host-only check (on a GPU box the spawned ranks would run the whole bench)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--unet", "tiny"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    # rank 0 stops at the device probe; the elastic agent then ends rank 1 and lists both ranks in its report
    assert "bench.py needs an MI355X" in p.stderr and "local_rank: 1" in p.stderr and "local_rank: 0" in p.stderr, p.stderr[-1500:]


def test_random_exposure_adjust_is_hot_path_subset_only():
    """Only the uint16 quantiser of the reference class is in scope; the training-time augmentation
    members raise instead of computing on the host."""
    from gm_diffusion import RandomExposureAdjust
    from gm_diffusion._native import HipExtensionError
    from gm_diffusion.stage1.augmentations import OutOfScopeError

    aug = RandomExposureAdjust(gamma=2.2, prob=1.0)
    with pytest.raises(OutOfScopeError):
        aug(torch.zeros(3, 8, 8))
    for call in (aug.hdr_to_ldr, RandomExposureAdjust.sample_camera_curve, RandomExposureAdjust.apply_inv_sigmoid_curve):
        with pytest.raises(OutOfScopeError):  # instance and class access behave the same
            call()
    with pytest.raises(OutOfScopeError):
        aug.gamma
    assert not hasattr(aug, "gamma") and not hasattr(aug, "exposure_levels")  # OutOfScopeError is an AttributeError too
    with pytest.raises(NotImplementedError):
        aug.prob
    with pytest.raises(HipExtensionError):  # the in-scope member is a HIP kernel: host tensors are refused
        RandomExposureAdjust.discretize_to_uint16(torch.zeros(4))


def test_predrawn_scheduler_noise_is_capped_and_ordered():
    """DDPM on the fused path pre-draws the variance noise of every step from a CPU generator, SDR before GM in each
    iteration (stable_diffusion_dual_unet.py:1077, 1093); above PREDRAW_NOISE_BYTES it declines and the loop draws per step."""
    from gm_diffusion.components import DDPMScheduler
    from gm_diffusion.components.image_processor import randn_tensor
    from gm_diffusion.pipelines import StableDiffusionGMPipeline as P

    s1 = DDPMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")
    s1.set_timesteps(4)
    s2 = DDPMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")
    s2.set_timesteps(4)
    ts = [int(t) for t in s1.timesteps]
    shape = (2, 4, 8, 8)
    pre = P._predraw_step_noise([s1, s2], ts, shape, torch.Generator().manual_seed(5), "cpu")
    g = torch.Generator().manual_seed(5)
    for i, t in enumerate(ts):
        for k, s_ in enumerate((s1, s2)):
            if s_.draws_noise(t):
                assert torch.equal(pre[k][i], randn_tensor(shape, generator=g, device="cpu", dtype=torch.float32)), (i, k)
            else:
                assert pre[k][i] is None
    old = P.PREDRAW_NOISE_BYTES
    try:
        P.PREDRAW_NOISE_BYTES = 4 * 2 * 4 * 8 * 8 * 3  # room for three draws only
        assert P._predraw_step_noise([s1, s2], ts, shape, torch.Generator().manual_seed(5), "cpu") is None
    finally:
        P.PREDRAW_NOISE_BYTES = old


def test_float32_contraction_mode_selection():
    """Which dtype code a contraction gets (hip_ops._contract_code): float32 tensors go to the matrix cores as three float16
    products (GMD_F32S, or GMD_F32SW for a pre-split weight) in mode 'split' when K is a multiple of 32, to the exact FMA
    kernels otherwise; 16-bit tensors never meet a pre-split weight; the mode switch validates its argument."""
    from gm_diffusion import hip_ops as ops
    from gm_diffusion._native import GMD_BF16, GMD_F16, GMD_F32, GMD_F32S, GMD_F32SW, HipExtensionError

    a, w = torch.zeros(4, 64), torch.zeros(8, 64)
    prev = ops.set_f32_mode("split")
    try:
        assert ops.f32_split() and ops._contract_code(a, w, 64) == GMD_F32S
        assert ops._contract_code(a[:, :48], w[:, :48], 48) == GMD_F32  # K % 32 != 0: exact kernel
        ws = torch.zeros(8, 64)
        ws._split = True
        assert ops._contract_code(a, ws, 64) == GMD_F32SW
        assert ops._contract_code(a.bfloat16(), w.bfloat16(), 64) == GMD_BF16 and ops._contract_code(a.half(), w.half(), 64) == GMD_F16
        with pytest.raises(HipExtensionError):
            ops._contract_code(a.half(), ws, 64)
        assert ops.set_f32_mode("exact") == "split" and not ops.f32_split()
        assert ops._contract_code(a, w, 64) == GMD_F32
        assert ops._contract_code(a, ws, 64) == GMD_F32SW  # a model prepared in split mode keeps its pre-split weights
        assert not ops.split_attention_ok(torch.float32, 40)
        ops.set_f32_mode("split")
        assert ops.split_attention_ok(torch.float32, 40) and ops.split_attention_ok(torch.float32, 64) and not ops.split_attention_ok(torch.float32, 32)
        assert not ops.split_attention_ok(torch.bfloat16, 40)
        with pytest.raises(HipExtensionError):
            ops.set_f32_mode("fast")
        with pytest.raises(HipExtensionError):
            ops.split_weights(torch.zeros(8, 64))  # host tensor: no CPU fallback
    finally:
        ops.set_f32_mode(prev)
