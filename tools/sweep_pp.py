#!/usr/bin/env python
"""Ping-pong GEMM kernel (gemm_pp_kernel, plan code 283) against the heuristic's choice on every linear / conv shape of the UNet at
batch 8 (SDR UNet with CFG at the bench workload) and batch 4 (GM UNet): device time inside a HIP graph (tools/bench_graph_ops.py)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = [("heuristic", ""), ("pp160", "256,160,283,1"), ("pp128", "256,128,283,1"), ("pp160k2", "256,160,283,2"), ("pp128k2", "256,128,283,2")]
if len(sys.argv) > 1 and sys.argv[1] == "--round3":
    # the whole heuristic against the round-3 plans (GMD_PP=0: no ping-pong, no loader/consumer kernel)
    variants = [("heuristic", ""), ("round3", "ENV:GMD_PP=0")]
elif len(sys.argv) > 1:
    variants = [("heuristic", "")] + [(v, v) for v in sys.argv[1:]]
for batch in (8, 4):
    tab = {}
    order = []
    for name, force in variants:
        cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_graph_ops.py"), "--batch", str(batch), "--only", "gemm,conv"]
        env = dict(os.environ)
        if force.startswith("ENV:"):
            k, v = force[4:].split("=")
            env[k] = v
        elif force:
            cmd += ["--force", force]
        out = subprocess.run(cmd, capture_output=True, text=True, env=env).stdout
        for line in out.splitlines():
            if line.startswith(("gemm", "conv")):
                parts = line.rsplit(None, 2)
                key = parts[0].strip()
                if key not in tab:
                    tab[key] = {}
                    order.append(key)
                tab[key][name] = float(parts[1])
    print(f"batch {batch}: us per launch")
    print(f"{'op':58s} " + " ".join(f"{n:>10s}" for n, _ in variants) + "   best")
    for key in order:
        row = tab[key]
        vals = [row.get(n, float('nan')) for n, _ in variants]
        best = min((v, n) for v, (n, _) in zip(vals, variants) if v == v)
        print(f"{key:58s} " + " ".join(f"{v:10.1f}" for v in vals) + f"   {best[1]} ({(row['heuristic'] / best[0] - 1) * 100:+.0f}%)", flush=True)
    for n, _ in variants:
        print(f"   sum {n}: {sum(tab[k].get(n, float('nan')) for k in order if 'vT' not in k):.1f} us (without the batched V^T rows)")
