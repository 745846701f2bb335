#!/usr/bin/env python
"""
bench.py -- HDR images/sec of the Stage-3 hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic prompts:
  dual-UNet denoising loop (50 PNDM steps -> 51 iterations; SDR UNet on 2B samples with CFG +
  GM UNet on B samples per iteration, fused HIP latent step) -> 2 VAE decodes -> fused HDR tail
  (Eq. 1, u8/u16 quantisation).  Text encoding is excluded (embeddings are inputs), weights are
  synthetic (no checkpoint exists offline) -- BASELINE.md §2/§3.

N=1 workload = BASELINE.json configs[1]: SD-v1-5 dual-UNet, 512x512, 50 PNDM steps, bf16, batch 4.
N>1: one rank per GPU over RCCL, weak scaling at `--batch` prompts per rank (or `--global-batch G` = G/N per rank:
`--gpus 8 --global-batch 64` is BASELINE.json configs[2]); rank 0 builds the full-batch embeddings / latents and
broadcasts them, every rank denoises its slice.  Launched either by `torch.distributed.run` (RANK / WORLD_SIZE in the
environment) or bare as `python bench.py --gpus N`: the parent then starts the N ranks itself as child processes BEFORE
anything touches the GPU and relays rank 0's line.

Prints ONE JSON line on rank 0 (contract in the task description), with
  "roofline":     the kind with the largest measured time: algorithmic FLOPs (or bytes) / HIP-event time,
  "kernels":      the same per kind, each with its fraction of the MFMA (2.5 PFLOP/s bf16) or HBM (8 TB/s) peak,
  "kernels_alone_plans": the MFMA kinds of the same step under the launch-by-launch plan family (a kernel that has the chip to itself),
  "cpu_baseline": the CPU oracle (a port, NOT diffusers) timed on this box's host cores on a bounded sample,
  "latent_rms_vs_f32": drift of the benchmarked precision against the float32 HIP path on a short fixed run.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

BF16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
F32_VECTOR_PEAK_TFLOPS = 157.3   # float32 parity path (FMA kernels)
HBM_PEAK_GBPS = 8000.0
MFMA_KINDS = ("conv3x3", "gemm_nt", "attention")
KIND_KERNEL = {
    "conv3x3": "gmd_conv3x3 (gemm_pp_kernel / gemm_ring_kernel<CONV=true> + splitk_reduce: implicit-GEMM conv3x3, 49.6% of UNet / 96.8% of VAE FLOPs; "
               "float32: gemm_split_kernel, three float16 MFMA passes)",
    "gemm_nt": "gmd_gemm_nt / gmd_ff_geglu_fused (gemm_pp_kernel / gemm_ring_kernel / gemm_bf16_kernel<CONV=false> + splitk_reduce, ff_fused_kernel: Linear / "
               "conv1x1 / GEGLU feed-forward; float32: gemm_split_kernel)",
    "attention": "gmd_attention (attn40_kernel / attn_fwd_kernel<D>: fused QK^T + softmax + PV, algorithmic head dim; float32: attn_split_kernel)",
    "groupnorm": "gmd_groupnorm_fused / gmd_groupnorm_split (GroupNorm + SiLU)",
    "layernorm": "gmd_layernorm",
    "concat": "gmd_concat_channels (skip connections)",
    "hdr_tail": "gmd_hdr_tail (denorm/clamp + u8 + Eq. 1 + /(qmax+1) + u16)",
}
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r05_pmc_traffic.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4, help="prompts per GPU (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0, help="total prompts over all GPUs (overrides --batch; must divide by the GPU count)")
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--inference-steps", type=int, default=50)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"],
                    help="bf16 = BASELINE.json's metric; f16 = the reference's own half type, same kernels (side line); f32 = the parity path")
    ap.add_argument("--unet", default="sd15", choices=["sd15", "tiny", "sdxl"],
                    help="tiny = structural smoke config (not a valid bench); sdxl = SDXL-base-width dual UNets with text_time conditioning "
                         "(BASELINE.json configs[4] shape; an extension -- the reference has no SDXL path; use with --res 1024 --inference-steps 30)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scheduler", default="pndm", choices=["pndm", "dpm++", "ddpm"],
                    help="pndm = BASELINE.json's metric; dpm++ / ddpm = the schedulers the reference's scripts construct (side measurements)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="run the GM UNet on the SDR stream instead of a second HIP stream")
    ap.add_argument("--no-graphs", action="store_true", help="launch every kernel eagerly instead of replaying captured HIP graphs")
    ap.add_argument("--no-drift", action="store_true", help="skip the short bf16-vs-float32 drift measurement")
    ap.add_argument("--no-tolerance-path", action="store_true",
                    help="skip timing the float32 (three float16 MFMA passes) pipeline -- the path inside the 1e-3 gate -- beside the headline")
    ap.add_argument("--tolerance-steps", type=int, default=2, help="timed full-workload steps of the tolerance path")
    ap.add_argument("--drift-steps", type=int, default=10)
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads of the cpu_baseline leg (a 1-GPU box owns 16 cores)")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    ap.add_argument("--checksum", action="store_true", help="add sha256 of the u16 HDR codes of the last step to the line (sharding tests)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# bare `python bench.py --gpus N`: start the ranks from a parent that never initialises the GPU
# ------------------------------------------------------------------------------------------------
def spawn_ranks(n):
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------
# CPU baseline (the oracle, test infrastructure: only this leg imports it)
# ------------------------------------------------------------------------------------------------
def cpu_baseline(res, steps, cores_hint=None, config1=True):
    """Time the CPU oracle on a bounded sample of the bench workload: ONE SD-1.5 UNet evaluation (batch 1) and ONE VAE
    decode at the bench resolution, float32, all host cores; extrapolate to one HDR image = (steps+1) x (2 UNet-4ch +
    1 UNet-8ch) + 2 decodes (the 8-channel UNet differs only in conv_in: timed as the 4-channel one).  Beside it BASELINE.json
    configs[0] (single 8-channel UNet, 1 prompt, 256x256, 10 PNDM steps, 2 decodes + Eq. 1) is timed IN FULL."""
    import torch

    from oracle import fixtures, pipelines as opipe, schedulers as osched

    cores = min(cores_hint or 16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    h = res // 8
    out = {}
    with torch.no_grad():
        unet = fixtures.build_unet("sd15", 8)
        x = torch.randn(1, 8, h, h)
        ctx = torch.randn(1, 77, 768)
        t0 = time.perf_counter()
        unet(x, torch.tensor(501), encoder_hidden_states=ctx)
        t_unet = time.perf_counter() - t0
        vae = fixtures.build_vae("sd15")
        t0 = time.perf_counter()
        vae.decode(torch.randn(1, 4, h, h))
        t_vae = time.perf_counter() - t0
        if config1:
            pos, neg, lat = fixtures.make_inputs(1, 32, 32)
            sdr_lat = torch.randn(1, 4, 32, 32, generator=torch.Generator().manual_seed(7)) * 0.7
            t0 = time.perf_counter()
            gm = opipe.gm_loop(unet, osched.PNDMScheduler(), sdr_lat, pos, neg, lat, num_inference_steps=10, guidance_scale=7.5)
            opipe.decode_tail(vae, sdr_lat, gm, qmax=99)
            t_c1 = time.perf_counter() - t0
            out["config1_full"] = {"seconds": round(t_c1, 2), "value": round(1.0 / t_c1, 5), "unit": "HDR images/s",
                                   "what": "BASELINE.json configs[0] timed in full: single 8-ch SD-1.5 UNet, 1 prompt, 256x256, 10 PNDM steps "
                                           "(11 iterations at CFG batch 2), 2 VAE decodes + Eq. 1, float32"}
    per_image = (steps + 1) * 3 * t_unet + 2 * t_vae
    out.update({
        "value": 1.0 / per_image, "unit": "HDR images/s", "cores": cores, "kind": "port",
        "sample": f"CPU oracle (pure-torch fp32 restatement, NOT diffusers): 1 SD-1.5 UNet eval (batch 1, {h}x{h} latent) = "
                  f"{t_unet:.2f}s + 1 VAE decode {res}x{res} = {t_vae:.2f}s; extrapolated to {(steps + 1) * 3} UNet evals + 2 decodes per image",
    })
    return out


def main():
    a = parse()
    if a.cpu_baseline_only:
        print(json.dumps(cpu_baseline(a.res, a.inference_steps, a.cpu_threads)))
        return
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))  # nothing above imported torch.cuda or made a HIP call

    import hashlib

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path is hand-written HIP with no CPU fallback")
    torch.cuda.set_device(local_rank)
    # torchrun exports OMP_NUM_THREADS=1: the synthetic-weight set-up (1.8 G random floats per rank) would crawl on one thread
    torch.set_num_threads(max(1, min(16, (os.cpu_count() or 8) // max(world, 1))))
    dev = torch.device("cuda", local_rank)
    # GMD_BENCH_FORCE_DIST=1 runs the RCCL path (process group, broadcasts, barrier, max-reduce) with a single rank too:
    # the only way to exercise it on a one-GPU box (tests/test_distributed_gpu.py)
    use_dist = world > 1 or (os.environ.get("GMD_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if a.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {a.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    from gm_diffusion import distributed as gdist, hdr, profiling
    from gm_diffusion.components import AutoencoderKL, DDPMScheduler, DPMSolverMultistepScheduler, PNDMScheduler, UNet2DConditionModel
    from gm_diffusion.pipelines import StableDiffusionDualUNetPipeline

    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
    tiny = a.unet == "tiny"
    ucfg = dict(block_out_channels=(64, 128, 128, 128), cross_attention_dim=64, attention_head_dim=2, norm_num_groups=8) if tiny else {}
    if a.unet == "sdxl":
        from gm_diffusion.components.unet_2d_condition import SDXL_UNET_CONFIG

        ucfg = dict(SDXL_UNET_CONFIG)
    vcfg = dict(block_out_channels=(64, 64, 128, 128), norm_num_groups=8) if tiny else {}

    def make_sched():
        kw = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1)
        if a.scheduler == "dpm++":
            return DPMSolverMultistepScheduler(timestep_spacing="leading", **kw)
        if a.scheduler == "ddpm":
            return DDPMScheduler(clip_sample=False, **kw)
        return PNDMScheduler(num_train_timesteps=1000, skip_prk_steps=True, set_alpha_to_one=False, **kw)

    def make_pipe(u, g, v):
        p = StableDiffusionDualUNetPipeline(vae=v, text_encoder=None, tokenizer=None, unet=u, gm_unet=g, scheduler=make_sched(),
                                            safety_checker=None, feature_extractor=None, requires_safety_checker=False)
        p.set_progress_bar_config(disable=True)
        p.overlap_streams = not a.no_overlap
        p.use_hip_graphs = not a.no_graphs
        return p

    t_build = time.perf_counter()
    # synthetic weights are drawn by the DEVICE generator (same seed -> the same weights on every rank): N ranks each drawing
    # 1.8 G host randoms on cpu_count/N threads would be the longest phase of a multi-GPU run, in front of the first collective
    unet = UNet2DConditionModel(in_channels=4, **ucfg).init_random(1234, device=dev).to(dev, dtype)
    gm_unet = UNet2DConditionModel(in_channels=8, **ucfg).init_random(1238, device=dev).to(dev, dtype)
    vae = AutoencoderKL(**vcfg).init_random(1334, device=dev).to(dev, dtype)
    pipe = make_pipe(unet, gm_unet, vae)

    if a.global_batch:
        if a.global_batch % world:
            raise SystemExit(f"--global-batch {a.global_batch} does not divide over {world} GPUs")
        B = a.global_batch // world
    else:
        B = a.batch
    total = B * world
    cross = unet.config.cross_attention_dim
    h = a.res // 8
    pos = neg = lat = None
    if rank == 0:  # rank 0 stands in for the text encoder: full-batch hidden states + full-batch noise
        ge = torch.Generator("cpu").manual_seed(1)
        pos = torch.randn(total, 77, cross, generator=ge)
        neg = torch.randn(total, 77, cross, generator=ge)
        lat = torch.randn(total, 4, h, h, generator=torch.Generator("cpu").manual_seed(42))
    # the text hidden states travel in the model dtype (north star: "RCCL broadcast of text-encoder hidden states"; 15 MB at
    # batch 64 in bf16), rounded on rank 0 exactly as the UNet's prepare_context would round them; the latents stay float32
    pos, neg, lat, (lo, hi) = gdist.shard_prompt_batch(pos, neg, lat, total, (77, cross), (4, h, h), dtype, dev, force=use_dist)
    pos, neg, lat = pos.to(dev), neg.to(dev), lat.to(dev)
    added = None
    if a.unet == "sdxl":  # pooled text embeddings + micro-conditioning: seeded, generated alike on every rank, sliced like the prompts
        ga = torch.Generator("cpu").manual_seed(2)
        te, nte = torch.randn(total, 1280, generator=ga), torch.randn(total, 1280, generator=ga)
        ids = torch.tensor([[float(a.res), float(a.res), 0.0, 0.0, float(a.res), float(a.res)]] * total)
        added = dict(text_embeds=te[lo:hi].to(dev), negative_text_embeds=nte[lo:hi].to(dev), time_ids=ids[lo:hi].to(dev))
    unet._ensure(); gm_unet._ensure(); vae._ensure()
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    gen_seed = 1234

    def step(p=pipe, v=vae, pe=pos, ne=neg, la=lat, n=a.inference_steps):
        kw = {}
        if a.scheduler == "ddpm":  # stochastic scheduler: one CPU generator shared by both scheduler steps (dual.py:1015)
            kw["generator"] = torch.Generator("cpu").manual_seed(gen_seed)
        if added is not None:
            kw["added_cond_kwargs"] = {k: v[: pe.shape[0]] for k, v in added.items()}
        sdr, gm = p(prompt_embeds=pe, negative_prompt_embeds=ne, latents=la, height=a.res, width=a.res,
                    num_inference_steps=n, guidance_scale=7.5, output_type="latent", **kw)
        return hdr.decode_to_hdr(v, sdr, gm, qmax=99.0, want=("sdr_u8", "gm_u8", "hdr", "hdr_u16")), sdr, gm

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        out, _, _ = step()
    fence()
    # Timed region: the product path as shipped (each UNet forward replayed from a captured HIP graph), no
    # instrumentation.  HIP events cannot bracket kernels inside a graph replay, so the per-kernel timing that feeds
    # `roofline` / `kernels` comes from ONE extra step of the same workload run eagerly (same kernels, same launch
    # parameters) right after the timed region, with torch.cuda.Event pairs on the launch stream around every launch.
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out, _, _ = step()
    fence()
    elapsed = time.perf_counter() - t0
    timer = None
    timer_alone = None
    path_diff = None
    if not a.no_kernel_timing and rank == 0:
        pipe.overlap_streams = False  # one stream: an event pair then brackets exactly one kernel running alone
        if hasattr(pipe, "co_run_plans"):
            pipe.co_run_plans = not a.no_overlap  # ... and the SAME launch plans as the timed region (its co-running family)
        # An event pair brackets a kernel only while the GPU has a backlog: when the GPU waits for the host (~33 k eager launches +
        # 66 k event records per step, host time ~ GPU time), elapsed(e0, e1) also holds the wait for the next launch, and the
        # short, numerous gemm_nt launches read 30 % too long on a box with a slow host (round 3: 36.8 vs 28.3 us average).  So
        # the host gets a head start: a dry instrumented step measures the host's enqueue time, then the stream is blocked by a
        # spinning kernel for that long (capped) while the host enqueues the measured step behind it.
        profiling.set_timer(profiling.KernelTimer())
        torch.cuda.synchronize()
        t_h = time.perf_counter()
        step()
        host_enqueue_s = time.perf_counter() - t_h
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.cuda._sleep(20_000_000); e1.record()
        torch.cuda.synchronize()
        ticks_per_ms = 20_000_000 / max(e0.elapsed_time(e1), 1e-3)
        host_lead_ms = min(host_enqueue_s * 1e3, 4000.0)
        timer = profiling.KernelTimer()
        profiling.set_timer(timer)  # an active timer makes the pipeline take the eager (non-graph) path
        torch.cuda._sleep(int(host_lead_ms * ticks_per_ms))
        out_eager, _, _ = step()
        torch.cuda.synchronize()
        profiling.set_timer(None)
        # The timed region launches the CO-RUNNING plan family (the two forwards share the chip: fewest L2 -> LDS bytes per product,
        # K slices reduced inside the kernel): faster on the wall, slower launch by launch when a kernel has the chip to itself, which
        # is how this instrumented step runs it.  For reference the same step under the launch-by-launch family (what every
        # single-stream caller of the C ABI gets) is timed too: `kernels_alone_plans`.
        if hasattr(pipe, "co_run_plans") and not a.no_overlap:
            pipe.co_run_plans = False
            timer_alone = profiling.KernelTimer()
            profiling.set_timer(timer_alone)
            torch.cuda._sleep(int(host_lead_ms * ticks_per_ms))
            step()
            torch.cuda.synchronize()
            profiling.set_timer(None)
        # the shipped path (graph replay, two streams) and this eager single-stream step run the same kernels on the same
        # inputs: their HDR images must be bit-identical -- a full-size guard against cross-stream races
        path_diff = float((out_eager["hdr"] - out["hdr"]).abs().max().item())
        pipe.overlap_streams = not a.no_overlap
        if hasattr(pipe, "co_run_plans"):
            pipe.co_run_plans = None
    if use_dist:
        dist.barrier()
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    finite = bool(torch.isfinite(out["hdr"]).all().item())

    # The float32 pipeline on the matrix cores (every float32 product as three float16 MFMA passes: csrc/gemm_split.hip,
    # attention_split.hip) is the path INSIDE the north star's 1e-3 latent-RMS gate (tests/test_pipeline_gpu.py: ~1e-5 against
    # the CPU oracle at full SD-1.5 width).  It is timed here on the full workload beside the headline (`tolerance_path`), and
    # it is the reference of the drift of the benchmarked 16-bit precision on a SHORT fixed run (1 prompt, `--drift-steps` PNDM
    # steps), absolute and relative to the latent RMS.  Its own distance from the exact float32 FMA kernels is measured on the
    # same short run.
    drift = None
    tol_path = None
    failed_legs = []
    # (single-GPU runs only, like cpu_baseline: in a multi-GPU run the other ranks would wait at the final barrier for rank 0)
    want_tol = not a.no_tolerance_path and a.unet == "sd15" and a.dtype in ("bf16", "f16") and world == 1
    want_drift = not a.no_drift and a.unet == "sd15" and a.dtype in ("bf16", "f16") and world == 1
    if rank == 0 and (want_tol or want_drift):
        from gm_diffusion import hip_ops

        rms = lambda x, y: float(((x.double() - y.double()) ** 2).mean().sqrt().item())
        prev_mode = hip_ops.set_f32_mode("split")
        f_pipe = f_unet = f_gm = f_vae = e_pipe = e_unet = e_gm = None
        pe1, ne1, la1 = pos[:1].contiguous(), neg[:1].contiguous(), lat[:1].contiguous()
        s_f = g_f = None
        lat_rms = None
        # three legs, each reported (and failing) on its own: a broken leg puts {"error": ...} under ITS key, the line is still
        # printed, and the process exits non-zero (failed_legs) -- a bench whose tolerance path is broken must not look green
        try:
            f_unet = UNet2DConditionModel(in_channels=4, **ucfg).load_state_dict(unet.state_dict()).to(dev, torch.float32)
            f_gm = UNet2DConditionModel(in_channels=8, **ucfg).load_state_dict(gm_unet.state_dict()).to(dev, torch.float32)
            f_vae = AutoencoderKL(**vcfg).load_state_dict(vae.state_dict()).to(dev, torch.float32)
            f_pipe = make_pipe(f_unet, f_gm, f_vae)
            _, s_f, g_f = step(f_pipe, f_vae, pe1, ne1, la1, a.drift_steps)
            lat_rms = float(s_f.double().pow(2).mean().sqrt().item())
        except Exception as e:  # pragma: no cover
            failed_legs.append("float32 reference run")
            if want_drift:
                drift = {"error": repr(e)}
            if want_tol:
                tol_path = {"error": repr(e)}
        if want_drift and drift is None:
            try:
                _, s_b, g_b = step(pipe, vae, pe1, ne1, la1, a.drift_steps)
                d_s, d_g = rms(s_b, s_f), rms(g_b, g_f)
                drift = {"sdr": round(d_s, 6), "gm": round(d_g, 6), "latent_rms": round(lat_rms, 4),
                         "sdr_rel": float("%.3g" % (d_s / lat_rms)), "gm_rel": float("%.3g" % (d_g / float(g_f.double().pow(2).mean().sqrt().item()))),
                         "pndm_steps": a.drift_steps, "prompts": 1, "resolution": a.res,
                         "note": f"RMS difference of the final latents, {a.dtype} path vs the float32 HIP path on the matrix cores (same weights, "
                                 "seed, embeddings); *_rel = divided by the RMS of the reference latents (the synthetic weights blow the latents "
                                 "up); the north-star gate 1e-3 is absolute and is met by the float32 path: see tolerance_path.  Against the CPU "
                                 "oracle at this width, 50 steps: tests/test_northstar_gpu.py"}
            except Exception as e:  # pragma: no cover
                failed_legs.append("latent_rms_vs_f32")
                drift = {"error": repr(e)}
        if want_tol and tol_path is None:
            try:
                for _ in range(2):  # warm-up: graph capture of the float32 forwards at the full batch, then one replayed step
                    step(f_pipe, f_vae)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                tol_each = []
                for _ in range(a.tolerance_steps):
                    step(f_pipe, f_vae)
                    torch.cuda.synchronize()
                    tol_each.append(round((time.perf_counter() - t1) * 1e3 - sum(tol_each), 1))
                t_tol = time.perf_counter() - t1
                # the split path against the exact float32 FMA kernels, same short run (modules keep the mode they were placed
                # on the device under: components/unet_2d_condition.py::_in_own_f32_mode)
                hip_ops.set_f32_mode("exact")
                e_unet = UNet2DConditionModel(in_channels=4, **ucfg).load_state_dict(unet.state_dict()).to(dev, torch.float32)
                e_gm = UNet2DConditionModel(in_channels=8, **ucfg).load_state_dict(gm_unet.state_dict()).to(dev, torch.float32)
                e_pipe = make_pipe(e_unet, e_gm, f_vae)
                e_pipe.use_hip_graphs = False
                sdr_e, gm_e = e_pipe(prompt_embeds=pe1, negative_prompt_embeds=ne1, latents=la1, height=a.res, width=a.res,
                                     num_inference_steps=a.drift_steps, guidance_scale=7.5, output_type="latent")
                torch.cuda.synchronize()
                tol_path = {
                    "dtype": "f32 (float32 tensors; every contraction as three float16 MFMA passes, f16 hi + f16 lo operands, fp32 accumulate)",
                    "images_per_s": round(B * a.tolerance_steps / t_tol, 4), "ms_per_step": round(t_tol / a.tolerance_steps * 1e3, 1),
                    "steps": a.tolerance_steps, "ms_each": tol_each, "workload": "same as config.workload (full batch, all inference steps, 2 float32 VAE decodes + tail)",
                    "latent_rms_vs_f32": {"sdr": float("%.3g" % rms(s_f, sdr_e)), "gm": float("%.3g" % rms(g_f, gm_e)),
                                          "sdr_rel": float("%.3g" % (rms(s_f, sdr_e) / lat_rms)), "reference": "exact float32 FMA kernels (GMD_F32_MODE=exact)",
                                          "pndm_steps": a.drift_steps, "prompts": 1},
                    "gate": "north star: latent RMS <= 1e-3 vs the float32 CPU reference.  tests/test_northstar_gpu.py holds THIS pipeline (both "
                            "SD-1.5-width UNets, 512x512, 50 PNDM steps, graphs + two streams) to it against the CPU oracle's committed "
                            "fixture, per recorded iteration and at the end: 1.2e-5 (SDR) / 8.7e-6 (GM) on MI355X",
                }
            except Exception as e:  # pragma: no cover
                failed_legs.append("tolerance_path")
                tol_path = {"error": repr(e)}
        hip_ops.set_f32_mode(prev_mode)
        del f_pipe, f_unet, f_gm, f_vae, e_pipe, e_unet, e_gm
        torch.cuda.empty_cache()

    if rank == 0:
        roof = None
        kernels = {}
        kernels_alone = {}
        if timer is not None:
            full = timer.summary()
            # f16 MFMA = bf16 rate; float32 runs on the same matrix cores in three float16 passes (algorithmic FLOPs are counted
            # once, so a perfect three-pass kernel would read 1/3) unless GMD_F32_MODE=exact selects the vector FMA kernels
            from gm_diffusion import hip_ops as _ops
            mfma_peak = BF16_DENSE_PEAK_TFLOPS if (a.dtype in ("bf16", "f16") or _ops.f32_split()) else F32_VECTOR_PEAK_TFLOPS
            all_ms = sum(v["ms"] for v in full.values())
            for k, v in full.items():
                e = {"launches": v["launches"], "ms": round(v["ms"], 3), "avg_us": round(v["avg_us"], 2), "share": round(v["ms"] / all_ms, 3)}
                if k in MFMA_KINDS:
                    e.update(bound="mfma", tflops=round(v["tflops"], 2), frac=round(v["tflops"] / mfma_peak, 4))
                else:
                    e.update(bound="hbm", gbps=round(v["gbps"], 1), frac=round(v["gbps"] / HBM_PEAK_GBPS, 4))
                kernels[k] = e
            if timer_alone is not None:
                for k, v in timer_alone.summary().items():
                    if k in MFMA_KINDS:
                        kernels_alone[k] = {"launches": v["launches"], "avg_us": round(v["avg_us"], 2), "tflops": round(v["tflops"], 2),
                                            "frac": round(v["tflops"] / mfma_peak, 4)}
            dk = max(full, key=lambda k: full[k]["ms"])  # the kind with the largest measured time
            dom = full[dk]
            traffic, traffic_note = None, "no PMC record for this kind under profiles/"
            try:
                with open(PMC_TRAFFIC_FILE) as f:
                    rec = json.load(f).get(dk)
                if rec:
                    traffic = rec["hbm_bytes_per_launch"]
                    traffic_note = rec["note"]
            except (OSError, ValueError, KeyError):
                pass
            mf = dk in MFMA_KINDS
            roof = {"kernel": KIND_KERNEL.get(dk, dk), "kind": dk, "bound": "mfma" if mf else "hbm",
                    "achieved": round(dom["tflops"] if mf else dom["gbps"], 2), "peak": mfma_peak if mf else HBM_PEAK_GBPS,
                    "unit": "TFLOP/s" if mf else "GB/s",
                    "frac": round((dom["tflops"] / mfma_peak) if mf else (dom["gbps"] / HBM_PEAK_GBPS), 4),
                    "traffic": traffic, "traffic_note": traffic_note,
                    "launches": dom["launches"], "avg_launch_us": round(dom["avg_us"], 2),
                    ("flops_per_launch_avg" if mf else "bytes_per_launch_avg"): round((dom["flops"] if mf else dom["bytes"]) / dom["launches"]),
                    "measured": "HIP events around every launch of this kind in one extra eager, single-stream step after the timed "
                                "region (the timed region replays HIP graphs on two streams, which events cannot enter); chosen as the "
                                "kind with the largest measured time over ALL instrumented kinds; the stream is held by a spinning "
                                "kernel for host_lead_ms while the host enqueues the step, so that no event pair includes a wait for the host",
                    "host_enqueue_ms": round(host_enqueue_s * 1e3, 1), "host_lead_ms": round(host_lead_ms, 1),
                    "share_of_instrumented_kernel_time": round(dom["ms"] / all_ms, 3)}
        is_metric = a.scheduler == "pndm" and a.inference_steps == 50 and a.res == 512 and a.dtype == "bf16" and a.unet == "sd15"
        res = {
            "metric": ("HDR images/sec @ 512x512, 50 PNDM steps, dual-UNet" if is_metric
                       else f"HDR images/sec @ {a.res}x{a.res}, {a.inference_steps} {a.scheduler} steps, {a.unet} dual-UNet, {a.dtype} [not the BASELINE metric]"),
            "value": round(total * a.steps / elapsed, 4),
            "unit": "HDR images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"{'SDXL-base-width' if a.unet == 'sdxl' else 'SD-v1-5'} dual-UNet (SDR 4ch + GM 8ch) {a.res}x{a.res}, {a.inference_steps} {a.scheduler.upper()} steps "
                                   f"({a.inference_steps + (1 if a.scheduler == 'pndm' else 0)} iterations), CFG 7.5, batch {B}/GPU, 2 VAE decodes + Eq.1 HDR tail"
                                   + (" [TINY smoke config - not a valid bench]" if tiny else ""),
                       "global_batch": total, "per_gpu_batch": B, "resolution": a.res, "inference_steps": a.inference_steps,
                       "parallelism": f"prompt-batch sharding x{world}, RCCL broadcast of text hidden states + latents",
                       "rccl_world_size": dist.get_world_size() if use_dist else None},
            "outputs_finite": finite, "graph_vs_eager_max_abs_diff": path_diff, "setup_s": round(t_build, 1),
            "latent_rms_vs_f32": drift, "tolerance_path": tol_path, "kernels": kernels, "kernels_alone_plans": kernels_alone or None, "roofline": roof,
        }
        if a.checksum:
            res["output_sha256"] = hashlib.sha256(out["hdr_u16"].cpu().numpy().tobytes()).hexdigest()
        if not a.no_cpu_baseline and world == 1:
            try:
                res["cpu_baseline"] = cpu_baseline(a.res, a.inference_steps, a.cpu_threads)
            except Exception as e:  # pragma: no cover
                res["cpu_baseline"] = {"error": repr(e)}
        if failed_legs:
            res["failed_legs"] = failed_legs
        print(json.dumps(res))
        sys.stdout.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if failed_legs:
        print(f"[bench] FAILED legs: {failed_legs} (see the \"error\" entries of the line above)", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
