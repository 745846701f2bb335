#!/usr/bin/env python
"""Fold the raw PMC lines of tools/collect_pmc_traffic_r5.sh (FETCH_SIZE / WRITE_SIZE per shape, KB) and the shape census of one loop
iteration (tools/gemm_shape_census.py) into the record bench.py reads for `roofline.traffic`: per shape the HBM-side bytes per launch
(FETCH_SIZE x 2: MI355X_MICROARCH.md, gfx950 reports half the bytes of wide coalesced reads; WRITE_SIZE exact) against the algorithmic
bytes, and the launch- and time-weighted ratios over the covered shapes of the gemm_nt kind.
    pmc_traffic_json.py pmc_raw.txt census.jsonl > profiles/r05_pmc_traffic.json"""
import json, sys

raw, census = sys.argv[1], sys.argv[2]
cnt = {}
for l in open(raw):
    p = l.rstrip("\n").split("\t")
    if len(p) >= 4:
        cnt.setdefault(p[0], {})[p[1]] = (float(p[2]), p[3])
rows = [json.loads(l) for l in open(census) if l.startswith("{") and '"key"' in l]
kind_tot = {}
for r in rows:
    kind_tot[r["kind"]] = kind_tot.get(r["kind"], 0.0) + r["total_us"]


def census_of(kind, match):
    return [r for r in rows if r["kind"] == kind and match(r["key"])]


def gemm_entry(tag):
    t = tag.split()
    if t[0] == "ff_geglu_fused":
        M, C = int(t[1]), 320
        alg = 3 * M * C * 2 + (8 * C * C + 4 * C * C) * 2 + (8 * C + C) * 4
        cs = census_of("gemm_nt", lambda k: k[1] == "ff_geglu_fused" and k[0] == M)
        name = f"fused GEGLU feed-forward M={M} C=320 (ff_fused_kernel)"
    else:
        M, N, K, mode = int(t[1]), int(t[2]), int(t[3]), t[4]
        nout = N // 2 if mode == "geglu" else N
        alg = (M * K + N * K + M * nout + (M * N if mode == "res" else 0)) * 2 + N * 4
        cs = census_of("gemm_nt", lambda k: k[0] == M and k[1] == N and k[2] == K and k[3] == 1 and (k[4] == mode))
        name = f"M={M} N={N} K={K} {mode}"
    return name, alg, cs


out = {}
shapes = []
for tag, c in cnt.items():
    if not (tag.startswith("gemm") or tag.startswith("ff_")) or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    name, alg, cs = gemm_entry(tag)
    rd, wr = c["FETCH_SIZE"][0] * 1024 * 2, c["WRITE_SIZE"][0] * 1024
    launches = sum(r["launches"] for r in cs)
    t_us = sum(r["total_us"] for r in cs)
    shapes.append(dict(shape=name, kernel=c["FETCH_SIZE"][1].split("(")[0], launches_per_iteration=launches,
                       share_of_gemm_nt_time=round(t_us / kind_tot["gemm_nt"], 4), fetch_size_kb=c["FETCH_SIZE"][0], write_size_kb=c["WRITE_SIZE"][0],
                       read_bytes=rd, written_bytes=wr, hbm_bytes=rd + wr, algorithmic_bytes=alg, ratio=round((rd + wr) / alg, 3), time_us=t_us))
shapes.sort(key=lambda s: -s["share_of_gemm_nt_time"])
L = sum(s["launches_per_iteration"] for s in shapes)
T = sum(s["time_us"] for s in shapes)
hbm = sum(s["hbm_bytes"] * s["launches_per_iteration"] for s in shapes) / max(L, 1)
alg = sum(s["algorithmic_bytes"] * s["launches_per_iteration"] for s in shapes) / max(L, 1)
tw = sum(s["ratio"] * s["time_us"] for s in shapes) / max(T, 1e-9)
cover = T / kind_tot["gemm_nt"]
for s in shapes:
    s.pop("time_us")
out["gemm_nt"] = dict(
    kernel=f"launch-weighted over {len(shapes)} gemm_nt shapes of one loop iteration under the co-running plan family (tools/gemm_shape_census.py: "
           f"{100 * cover:.1f} % of the kind's time, the level-0 fused feed-forward included)",
    shape="see shapes[]", hbm_bytes_per_launch=int(hbm), algorithmic_bytes_per_launch=int(alg), ratio_launch_weighted=round(hbm / alg, 3),
    ratio_time_weighted=round(tw, 3), coverage_of_kind_time=round(cover, 3), shapes=shapes,
    note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) on tools/one_gemm.py / one_ff.py with GMD_ONE_FAMILY=1 "
         "(tools/collect_pmc_traffic_r5.sh, profiles/r05_pmc_raw.txt); FETCH_SIZE x2 (gfx950 wide-stream correction), WRITE_SIZE exact; per launch, "
         "weighted by launches per loop iteration.  Split launches (K slices reduced inside the kernel) count their accumulator fragments "
         "(sc1 stores and loads) in both directions.")
convs = []
for tag, c in cnt.items():
    if tag.startswith("conv") and "FETCH_SIZE" in c and "WRITE_SIZE" in c and "[" not in tag:
        B, H, W, ci, co = [int(v) for v in tag.split()[1:6]]
        alg = (B * H * W * ci + co * 9 * ci + B * H * W * co) * 2 + co * 4
        rd, wr = c["FETCH_SIZE"][0] * 1024 * 2, c["WRITE_SIZE"][0] * 1024
        e = dict(shape=f"conv3x3 B={B} {H}x{W} {ci}->{co} +bias", kernel=c["FETCH_SIZE"][1].split("(")[0], read_bytes=rd, written_bytes=wr,
                 hbm_bytes=rd + wr, algorithmic_bytes=alg, ratio=round((rd + wr) / alg, 3), read_ratio=round(rd / ((B * H * W * ci + co * 9 * ci) * 2), 3))
        for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT"):
            if k in c:
                e[k] = c[k][0]
        if "SQ_INSTS_MFMA" in e:
            e["valu_per_mfma"] = round(e["SQ_INSTS_VALU"] / e["SQ_INSTS_MFMA"], 2)
        convs.append(e)
out["conv3x3"] = dict(shapes=convs, shape=convs[0]["shape"] if convs else "", hbm_bytes_per_launch=int(convs[0]["hbm_bytes"]) if convs else 0,
                      kernel=convs[0]["kernel"] if convs else "", note="same passes; the dominant conv3x3 shape first")
print(json.dumps(out, indent=1))
