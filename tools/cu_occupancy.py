#!/usr/bin/env python
"""Per-CU occupancy of the shipped two-stream pipeline (BASELINE config 2: SD-1.5 dual UNet, 512x512, batch 4, bf16, captured graphs, GM
stream one step behind), from the per-workgroup trace of the DIAGNOSTIC library (csrc/wg_trace.h, tools/dbg/build_wgtrace.sh): every
workgroup of every kernel records the device clock at entry / exit of its wave 0, HW_ID (CU, SIMD, hardware queue) and XCC_ID.

Answers, for loop iterations in the middle of the run and per UNet block (block windows from the gmd_stamp table of the SDR forward):
  * fraction of CU-time with no workgroup resident / with workgroups of ONE stream / with workgroups of BOTH streams resident;
  * resident waves per CU (of 32 = 8 per SIMD);
  * per stream: time with NO workgroup of that stream anywhere on the chip (dispatch gaps between kernels) and with fewer than 64 CUs
    holding one (tails of a launch / launches that cannot fill the chip);
  * CU-time by kernel kind, and each kind's mean workgroup duration while co-running vs. with the streams serialised (--no-overlap pass).
Usage: cu_occupancy.py [--iters 10] [--batch 4] [--out file]      (loads tools/dbg/libgmd_wgtrace.so through GMD_LIB_OVERRIDE)"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GMD_LIB_OVERRIDE", os.path.join(ROOT, "tools", "dbg", "libgmd_wgtrace.so"))
for p in (ROOT, os.path.join(ROOT, "gm-diffusion_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from gm_diffusion._native import lib
from gm_diffusion.components import AutoencoderKL, PNDMScheduler, UNet2DConditionModel
from gm_diffusion.pipelines import StableDiffusionDualUNetPipeline

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10, help="loop iterations analysed (from iteration 8 on; at most 16)")
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--out", default="")
ap.add_argument("--dump", default="", help="save the raw records of the analysed window (npz)")
ap.add_argument("--pp-detail", action="store_true",
                help="(library built with -DGMD_PP_DIAG=1 as well) only report where the waves of gemm_pp_kernel spend their K loop WHILE CO-RUNNING")
a = ap.parse_args()
dev = torch.device("cuda", 0)
B, steps = a.batch, 50
IT0, IT1 = 8, 8 + a.iters
BLOCKS = ["down0 64x64", "down1 32x32", "down2 16x16", "down3 8x8", "mid 8x8", "up0 8x8", "up1 16x16", "up2 32x32", "up3 64x64", "out"]
KIND = {1: "gemm64", 2: "gemm_ring", 3: "gemm_pp", 4: "gemm_lc", 5: "conv_patch", 6: "conv_patch_cont", 7: "splitk_reduce", 8: "ff_fused", 9: "attn",
        10: "attn40", 11: "gn_partial", 12: "gn_apply_ws", 13: "gn_apply_cs", 14: "gn_fused", 15: "gn_fused_reg", 16: "gn_slab", 17: "layernorm",
        18: "layernorm_packed", 19: "concat", 20: "dup", 21: "pack", 22: "unpack", 23: "latent_step", 24: "temb", 25: "cast", 26: "stamp",
        27: "gn_finalize", 28: "gn_apply", 29: "cfg_ratio", 30: "gemm_f32", 31: "other"}


def kind_name(k):
    return ("conv:" if k & 64 else "") + KIND.get(k & 63, str(k & 63))


def build():
    unet = UNet2DConditionModel(in_channels=4).init_random(1234, device=dev).to(dev, torch.bfloat16)
    gm = UNet2DConditionModel(in_channels=8).init_random(1238, device=dev).to(dev, torch.bfloat16)
    vae = AutoencoderKL().init_random(1334, device=dev).to(dev, torch.bfloat16)
    sched = PNDMScheduler(num_train_timesteps=1000, skip_prk_steps=True, set_alpha_to_one=False, beta_start=0.00085, beta_end=0.012,
                          beta_schedule="scaled_linear", steps_offset=1)
    p = StableDiffusionDualUNetPipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, gm_unet=gm, scheduler=sched, safety_checker=None,
                                        feature_extractor=None, requires_safety_checker=False)
    p.set_progress_bar_config(disable=True)
    p.co_run_plans = True  # the serialised pass runs the SAME launches (the co-running plan family) as the shipped two-stream pass
    for m in (unet, gm):
        m._stamp_buf = torch.zeros(steps + 1, 16, dtype=torch.int64, device=dev)
        m._stamp_row = torch.zeros(1, dtype=torch.int32, device=dev)
    return p


SHARDS, CAP = 2048, 65536  # one record array per possible CU id, 65536 records each (4.3 GB on the device; ~2 k records per CU and iteration)
TRACE_STEPS = 25            # the traced call runs 26 loop iterations (the analysed ones are 8..17): the arrays hold that


def traced_run(p, overlap):
    p.overlap_streams = overlap
    ge = torch.Generator("cpu").manual_seed(1)
    pos, neg = torch.randn(B, 77, 768, generator=ge).to(dev), torch.randn(B, 77, 768, generator=ge).to(dev)
    lat = torch.randn(B, 4, 64, 64, generator=torch.Generator("cpu").manual_seed(42)).to(dev)
    kw = dict(prompt_embeds=pos, negative_prompt_embeds=neg, latents=lat, height=512, width=512, num_inference_steps=TRACE_STEPS, guidance_scale=7.5,
              output_type="latent")
    enable = lib().gmd_wg_trace_enable
    enable.argtypes = [ctypes.c_void_p]
    enable.restype = ctypes.c_int
    assert enable(None) == 0
    for _ in range(2):  # capture, then one warm replayed run, untraced
        p(**kw)
    torch.cuda.synchronize()
    t_untraced = (int(p.unet._stamp_buf[IT1, 0]) - int(p.unet._stamp_buf[IT0, 0])) / (IT1 - IT0) / 100
    ring = torch.zeros(16 + 16 * SHARDS + 4 * SHARDS * CAP, dtype=torch.int64, device=dev)
    ring[0] = SHARDS
    ring[1] = CAP
    torch.cuda.synchronize()
    assert enable(ring.data_ptr()) == 0
    p(**kw)
    torch.cuda.synchronize()
    assert enable(None) == 0
    counts = ring[16:16 + 16 * SHARDS:16].cpu().numpy()
    if counts.max() > CAP:
        print(f"# record arrays overflowed: up to {counts.max()} records on one CU, {CAP} kept")
    body = ring[16 + 16 * SHARDS:].view(SHARDS, CAP, 4)
    parts = [body[sh, :min(int(n), CAP)].cpu().numpy() for sh, n in enumerate(counts) if n > 0]
    rec = np.concatenate(parts).view(np.uint64)
    del ring, body
    s = p.unet._stamp_buf.cpu().numpy().astype(np.int64)
    g = p.gm_unet._stamp_buf.cpu().numpy().astype(np.int64)
    print(f"# {'two streams' if overlap else 'serialised'}: loop period untraced {t_untraced:.1f} us, traced {(s[IT1, 0] - s[IT0, 0]) / (IT1 - IT0) / 100:.1f} us "
          f"({len(rec)} workgroup records on {int((counts > 0).sum())} CUs)")
    return rec, s, g, t_untraced


def decode(rec):
    t0 = rec[:, 0].astype(np.int64)
    t1 = rec[:, 1].astype(np.int64)
    hw = (rec[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
    tag = (rec[:, 2] >> np.uint64(32)).astype(np.int64)
    block = (rec[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
    grid = (rec[:, 3] >> np.uint64(32)).astype(np.int64)
    d = dict(t0=t0, t1=t1, simd=(hw >> 4) & 3, pipe=(hw >> 6) & 3, cu=(hw >> 8) & 15, sh=(hw >> 12) & 1, se=(hw >> 13) & 7, queue=(hw >> 24) & 7,
             me=(hw >> 30) & 3, xcc=tag & 15, kind=(tag >> 8) & 255, waves=(tag >> 16) & 255, block=block, grid=grid)
    d["cuid"] = ((d["xcc"] * 8 + d["se"]) * 2 + d["sh"]) * 16 + d["cu"]
    d["q"] = (d["me"] * 4 + d["pipe"]) * 8 + d["queue"]
    return d


def stream_of_queues(d, s_tab, g_tab):
    """hardware queue -> 'sdr' / 'gm' through the stamp kernels' own trace records (their entry time is within a microsecond of the
    value they wrote into their UNet's table)."""
    st = d["kind"] == 26
    out = {}
    sv, gv = np.sort(s_tab[s_tab > 0]), np.sort(g_tab[g_tab > 0])
    for q in np.unique(d["q"]):
        m = st & (d["q"] == q)
        if not m.any():
            continue
        t = d["t0"][m]
        def near(v):
            i = np.clip(np.searchsorted(v, t), 1, len(v) - 1)
            return np.minimum(np.abs(v[i] - t), np.abs(v[i - 1] - t))
        ns, ng = (near(sv) < 200).mean() if len(sv) > 1 else 0, (near(gv) < 200).mean() if len(gv) > 1 else 0
        out[int(q)] = "sdr" if ns >= ng else "gm"
    return out


def sweep(t0, t1, w0, w1, weight=None):
    """total time inside [w0, w1] during which at least one of the intervals is open, and the integral of the open count (x weight)"""
    t0 = np.clip(t0, w0, w1); t1 = np.clip(t1, w0, w1)
    keep = t1 > t0
    t0, t1 = t0[keep], t1[keep]
    if len(t0) == 0:
        return 0, 0
    wt = np.ones(len(t0), dtype=np.int64) if weight is None else weight[keep]
    ev_t = np.concatenate([t0, t1]); ev_d = np.concatenate([np.ones(len(t0), dtype=np.int64), -np.ones(len(t1), dtype=np.int64)])
    ev_w = np.concatenate([wt, -wt])
    o = np.argsort(ev_t, kind="stable")
    ev_t, ev_d, ev_w = ev_t[o], ev_d[o], ev_w[o]
    cnt = np.cumsum(ev_d)[:-1]; cw = np.cumsum(ev_w)[:-1]
    dt = np.diff(ev_t)
    return int(dt[cnt > 0].sum()), int((dt * cw).sum())


def analyse(d, qmap, s_tab, g_tab, label, lines):
    P = lines.append
    strm = np.array([{"sdr": 0, "gm": 1}.get(qmap.get(int(q), "?"), 2) for q in d["q"]])
    W0, W1 = int(s_tab[IT0, 0]), int(s_tab[IT1, 0])
    niter = IT1 - IT0
    P(f"== {label}: iterations {IT0}..{IT1 - 1}; loop period {(W1 - W0) / niter / 100:.1f} us; window {(W1 - W0) / 100:.0f} us")
    inw = (d["t1"] > W0) & (d["t0"] < W1) & (d["kind"] != 26)
    cu_ids = np.unique(d["cuid"][inw])
    ncu = len(cu_ids)
    P(f"CUs seen: {ncu} (xcc {sorted(set(d['xcc'][inw].tolist()))}); hardware queues -> stream: {qmap}; workgroups in window: {int(inw.sum())} "
      f"({int((inw & (strm == 0)).sum())} SDR, {int((inw & (strm == 1)).sum())} GM, {int((inw & (strm == 2)).sum())} other)")
    order = np.argsort(d["cuid"][inw], kind="stable")
    idx = np.nonzero(inw)[0][order]
    bounds = np.searchsorted(d["cuid"][idx], cu_ids)
    bounds = np.append(bounds, len(idx))

    def window_stats(w0, w1):
        tot = dict(any=0, sdr=0, gm=0, both=0, waves=0)
        for c in range(ncu):
            ii = idx[bounds[c]:bounds[c + 1]]
            a0, a1, sm, wv = d["t0"][ii], d["t1"][ii], strm[ii], d["waves"][ii]
            anyt, wint = sweep(a0, a1, w0, w1, wv)
            st, _ = sweep(a0[sm == 0], a1[sm == 0], w0, w1)
            gt, _ = sweep(a0[sm == 1], a1[sm == 1], w0, w1)
            tot["any"] += anyt; tot["sdr"] += st; tot["gm"] += gt; tot["both"] += st + gt - anyt; tot["waves"] += wint
        T = (w1 - w0) * ncu
        return {k: v / T for k, v in tot.items()}

    ws = window_stats(W0, W1)
    P(f"CU-time over the window: idle {100 * (1 - ws['any']):.1f} %   one stream only {100 * (ws['any'] - ws['both']):.1f} % "
      f"(SDR only {100 * (ws['sdr'] - ws['both']):.1f} %, GM only {100 * (ws['gm'] - ws['both']):.1f} %)   both streams resident {100 * ws['both']:.1f} %   "
      f"resident waves per CU {ws['waves']:.1f} of 32")
    # per stream: chip-wide gaps / thin phases
    for si, nm in ((0, "SDR"), (1, "GM")):
        m = inw & (strm == si)
        if not m.any():
            continue
        anyt, _ = sweep(d["t0"][m], d["t1"][m], W0, W1)
        # CUs holding a workgroup of this stream over time: sweep on per-CU "open" transitions is costly; approximate with the count of
        # resident workgroups (>= 64 workgroups resident <=> at least 64 CU slots busy, exact for one-workgroup-per-CU kernels)
        t0c = np.clip(d["t0"][m], W0, W1); t1c = np.clip(d["t1"][m], W0, W1)
        ev_t = np.concatenate([t0c, t1c]); ev_d = np.concatenate([np.ones(len(t0c), dtype=np.int64), -np.ones(len(t1c), dtype=np.int64)])
        o = np.argsort(ev_t, kind="stable"); ev_t, ev_d = ev_t[o], ev_d[o]
        cnt = np.cumsum(ev_d)[:-1]; dt = np.diff(ev_t)
        thin = dt[(cnt > 0) & (cnt < 64)].sum()
        P(f"{nm} stream: no workgroup anywhere on the chip {100 * (1 - anyt / (W1 - W0)):.1f} % of the time (dispatch gaps), 1..63 workgroups resident "
          f"{100 * thin / (W1 - W0):.1f} % (tails / launches that cannot fill the chip), >= 64 resident {100 * dt[cnt >= 64].sum() / (W1 - W0):.1f} %")
    # per UNet block (SDR forward's block windows, averaged over the iterations)
    P(f"per block of the SDR forward (its block windows; the GM forward of the previous iteration runs beside it): us per iteration | idle | one stream | both | waves/CU")
    acc = {}
    for it in range(IT0, IT1):
        for k, nm in enumerate(BLOCKS):
            w0, w1 = int(s_tab[it, k]), int(s_tab[it, k + 1])
            if w1 <= w0:
                continue
            r = window_stats(w0, w1)
            e = acc.setdefault(nm, [0.0, 0.0, 0.0, 0.0, 0.0, 0])
            e[0] += (w1 - w0) / 100; e[1] += (1 - r["any"]) * (w1 - w0); e[2] += (r["any"] - r["both"]) * (w1 - w0); e[3] += r["both"] * (w1 - w0)
            e[4] += r["waves"] * (w1 - w0); e[5] += (w1 - w0)
    for nm in BLOCKS:
        if nm in acc:
            e = acc[nm]
            P(f"  {nm:14s} {e[0] / niter:8.1f} us | {100 * e[1] / e[5]:5.1f} % | {100 * e[2] / e[5]:5.1f} % | {100 * e[3] / e[5]:5.1f} % | {e[4] / e[5]:5.1f}")
    # between the end of the SDR forward and the start of the next (latent step, pack: the loop's serial tail)
    gap = np.mean([s_tab[it + 1, 0] - s_tab[it, 10] for it in range(IT0, IT1)]) / 100
    P(f"  between SDR forwards (latent step + pack + graph launch): {gap:.1f} us per iteration")
    # by kind
    P("CU-time by kernel kind (sum of workgroup durations / (CUs x window); > 100 % in total where workgroups share a CU) and mean workgroup duration:")
    rows = []
    for si, nm in ((0, "SDR"), (1, "GM")):
        for k in np.unique(d["kind"][inw & (strm == si)]):
            m = inw & (strm == si) & (d["kind"] == k)
            dur = (np.minimum(d["t1"][m], W1) - np.maximum(d["t0"][m], W0))
            rows.append((dur.sum() / ((W1 - W0) * ncu), nm, kind_name(int(k)), int(m.sum()) / niter, (d["t1"][m] - d["t0"][m]).mean() / 100,
                         float(np.median(d["waves"][m]))))
    for share, nm, kn, n, mean, wv in sorted(rows, reverse=True)[:28]:
        P(f"  {nm:4s} {kn:22s} {100 * share:6.2f} %   {n:9.0f} workgroups / iteration   mean {mean:8.2f} us   {wv:.0f} waves")
    return ws


def kind_durations(d, qmap, s_tab):
    strm = np.array([{"sdr": 0, "gm": 1}.get(qmap.get(int(q), "?"), 2) for q in d["q"]])
    W0, W1 = int(s_tab[IT0, 0]), int(s_tab[IT1, 0])
    inw = (d["t1"] > W0) & (d["t0"] < W1) & (d["kind"] != 26)
    out = {}
    for si, nm in ((0, "SDR"), (1, "GM")):
        for k in np.unique(d["kind"][inw & (strm == si)]):
            m = inw & (strm == si) & (d["kind"] == k)
            out[(nm, kind_name(int(k)))] = ((d["t1"][m] - d["t0"][m]).mean() / 100, int(m.sum()))
    return out


def pp_detail(rec, qmap, lines, label):
    """detail records of the stamped gemm_pp_kernel (kind | 0x80): per wave the cycle sums of its K-loop segments"""
    tag = (rec[:, 2] >> np.uint64(32)).astype(np.int64)
    kind = (tag >> 8) & 255
    m = ((kind & 0x80) != 0) & (((tag >> 16) & 255) != 255)
    r = rec[m]
    if len(r) == 0:
        return
    a0 = (r[:, 0] & np.uint64(0xffffffff)).astype(np.float64); a1 = (r[:, 0] >> np.uint64(32)).astype(np.float64)
    a2 = (r[:, 1] & np.uint64(0xffffffff)).astype(np.float64); a3 = (r[:, 1] >> np.uint64(32)).astype(np.float64)
    hw = (r[:, 2] & np.uint64(0xffffffff)).astype(np.int64); tg = tag[m]
    q = (((hw >> 30) & 3) * 4 + ((hw >> 6) & 3)) * 8 + ((hw >> 24) & 7)
    wid = (tg >> 16) & 255; conv = ((tg >> 8) & 64) != 0
    nk = (r[:, 3] & np.uint64(0xffff)).astype(np.int64); tn = ((r[:, 3] >> np.uint64(16)) & np.uint64(0xffff)).astype(np.int64)
    tot = (r[:, 3] >> np.uint64(32)).astype(np.float64)
    lines.append(f"== gemm_pp_kernel K-loop split, {label} (stamped build: every stamp adds ~40 cycles; shares, not times): shader cycles per K step and wave")
    lines.append("   stream kind  TN  K steps  waves | consumers: reads+wait  barrier(R)  mfma issue  barrier(C)  = loop | loaders: dma issue  barrier  vmcnt wait")
    for strm in ("sdr", "gm"):
        for cv in (False, True):
            for t in np.unique(tn):
                for k in np.unique(nk):
                    sel = (np.array([qmap.get(int(x), "?") for x in q]) == strm) & (conv == cv) & (tn == t) & (nk == k) & (nk > 0)
                    if sel.sum() < 512:
                        continue
                    c = sel & (wid < 8); l = sel & (wid >= 8)
                    f = lambda x, mm: (x[mm] / nk[mm]).mean() if mm.any() else float("nan")
                    lines.append(f"   {strm:4s} {'conv' if cv else 'gemm'}  {t:2d}  {k:7d}  {int(sel.sum()):6d} | {f(a0, c):8.0f} {f(a1, c):10.0f} {f(a2, c):10.0f} {f(a3, c):10.0f}  = {f(tot, c):6.0f} | "
                                 f"{f(a0, l):8.0f} {f(a1, l):8.0f} {f(a3, l):8.0f}")


def pp_phases(rec, lines, label):
    """phase records of gemm_pp_kernel workgroups (wave 0; wid field 0xFF), 100 MHz ticks: kernel entry -> K loop | K loop | barrier +
    in-kernel reduction | epilogue up to the last store issued"""
    tag = (rec[:, 2] >> np.uint64(32)).astype(np.int64)
    ph = rec[((((tag >> 8) & 255) & 0x80) != 0) & (((tag >> 16) & 255) == 255)]
    if len(ph) == 0:
        return
    f = lambda col, hi: ((ph[:, col] >> np.uint64(32)) if hi else (ph[:, col] & np.uint64(0xffffffff))).astype(np.int64)
    pro, loop, sync, epi = f(0, 0) / 100.0, f(0, 1) / 100.0, f(1, 0) / 100.0, f(1, 1) / 100.0
    ptag = f(2, 1)
    conv = ((ptag >> 8) & 64) != 0
    nk, tn = f(3, 0) & 0xffff, (f(3, 0) >> 16) & 0xffff
    g = f(3, 1)
    grid, ksl, res, fix = g & 0xfffff, (g >> 20) & 255, (g >> 28) & 1, (g >> 29) & 1
    lines.append(f"== gemm_pp_kernel workgroup phases, {label}: median us of wave 0 (entry -> K loop | K loop, stamped: ~10 % long | barrier + in-kernel reduction | epilogue to the last store issued)")
    lines.append("   kind  TN  K steps  grid  slices res fixup  workgroups |  prologue     loop     sync  epilogue |    sum")
    seen = {}
    for i in range(len(ph)):
        seen.setdefault((bool(conv[i]), int(tn[i]), int(nk[i]), int(grid[i]), int(ksl[i]), int(res[i]), int(fix[i])), []).append(i)
    for key in sorted(seen, key=lambda k: -len(seen[k]))[:30]:
        ii = np.array(seen[key])
        cv, t, k, gg, ks_, r_, fx = key
        m = [float(np.median(x[ii])) for x in (pro, loop, sync, epi)]
        lines.append(f"   {'conv' if cv else 'gemm'}  {t:2d}  {k:7d}  {gg:4d}  {ks_:6d} {r_:3d} {fx:5d}  {len(ii):10d} | {m[0]:8.2f} {m[1]:8.2f} {m[2]:8.2f} {m[3]:8.2f} | {sum(m):6.2f}")


def drop_detail(rec):
    kind = ((rec[:, 2] >> np.uint64(40)) & np.uint64(255)).astype(np.int64)
    return rec[(kind & 0x80) == 0]


lines = []
p = build()
rec, s_tab, g_tab, t_un = traced_run(p, True)
if a.pp_detail:
    d = decode(drop_detail(rec))
    qmap = stream_of_queues(d, s_tab, g_tab)
    pp_detail(rec, qmap, lines, "two streams (shipped)")
    pp_phases(rec, lines, "two streams (shipped)")
    rec2, s2, g2, _ = traced_run(p, False)
    q2 = {int(x): "sdr" for x in np.unique(decode(drop_detail(rec2))["q"])}
    pp_detail(rec2, q2, lines, "streams serialised (both forwards on one queue, listed as sdr)")
    txt = "\n".join(lines)
    print(txt)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    sys.exit(0)
d = decode(rec)
qmap = stream_of_queues(d, s_tab, g_tab)
lines.append(f"trace overhead: loop period {t_un:.1f} us untraced (same diagnostic library, trace pointer null) -> {(s_tab[IT1, 0] - s_tab[IT0, 0]) / (IT1 - IT0) / 100:.1f} us traced")
analyse(d, qmap, s_tab, g_tab, "two streams (shipped)", lines)
ov = kind_durations(d, qmap, s_tab)
if a.dump:
    W0, W1 = int(s_tab[IT0, 0]), int(s_tab[IT0 + 2, 0])
    m = (d["t1"] > W0) & (d["t0"] < W1)
    np.savez_compressed(a.dump, rec=rec[m], s_tab=s_tab, g_tab=g_tab)
del rec, d
rec, s_tab2, g_tab2, t_un2 = traced_run(p, False)
d2 = decode(rec)
# serialised: one queue carries both forwards; split by the stamp table (SDR forward window vs GM forward window)
qmap2 = stream_of_queues(d2, s_tab2, g_tab2)
lines.append("")
W0, W1 = int(s_tab2[IT0, 0]), int(s_tab2[IT1, 0])
inw = (d2["t1"] > W0) & (d2["t0"] < W1) & (d2["kind"] != 26)
cu_ids = np.unique(d2["cuid"][inw]); ncu = len(cu_ids)
tot_any = 0; tot_w = 0
for c in cu_ids:
    m = inw & (d2["cuid"] == c)
    at, wi = sweep(d2["t0"][m], d2["t1"][m], W0, W1, d2["waves"][m])
    tot_any += at; tot_w += wi
lines.append(f"== streams serialised (--no-overlap): loop period untraced {t_un2:.1f} us, traced {(W1 - W0) / (IT1 - IT0) / 100:.1f} us; CU-time idle {100 * (1 - tot_any / ((W1 - W0) * ncu)):.1f} %, "
             f"resident waves per CU {tot_w / ((W1 - W0) * ncu):.1f} of 32")
# mean workgroup duration per kind: co-running vs serialised.  In the serialised run both forwards share a queue: attribute by time window
in_sdr = np.zeros(len(d2["t0"]), dtype=bool)
for it in range(IT0, IT1):
    in_sdr |= (d2["t0"] >= s_tab2[it, 0]) & (d2["t0"] < s_tab2[it, 10])
se = {}
for nm, msk in (("SDR", in_sdr), ("GM", ~in_sdr)):
    for k in np.unique(d2["kind"][inw & msk]):
        m = inw & msk & (d2["kind"] == k)
        se[(nm, kind_name(int(k)))] = ((d2["t1"][m] - d2["t0"][m]).mean() / 100, int(m.sum()))
lines.append("mean workgroup duration by kind, us: co-running | serialised | ratio   (the stretch a workgroup suffers from the other stream's kernels)")
for key in sorted(ov, key=lambda k: -ov[k][0] * ov[k][1]):
    if key in se and ov[key][1] > 50:
        lines.append(f"  {key[0]:4s} {key[1]:22s} {ov[key][0]:8.2f} | {se[key][0]:8.2f} | x{ov[key][0] / se[key][0]:.2f}   ({ov[key][1] // (IT1 - IT0)} workgroups / iteration)")
txt = "\n".join(lines)
print(txt)
if a.out:
    open(a.out, "w").write(txt + "\n")
