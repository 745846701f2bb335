#!/usr/bin/env python
"""Interleaved A/B of bench.py variants on ONE box (the boxes differ by 1-2 %, so variants must be compared inside one lease):
each variant is an environment setting `NAME=VALUE[,NAME=VALUE...]`; the variants are run round-robin for `--rounds` rounds as
child processes and the per-variant ms_per_step values are printed with their median.
Usage: ab_bench.py [--rounds 3] [--steps 6] [--args "--no-overlap"] VAR1 VAR2 ...   (a variant "-" = no extra environment)"""
import argparse, json, os, statistics, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--args", default="")
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
res = {v: [] for v in a.variants}
for r in range(a.rounds):
    for v in a.variants:
        env = dict(os.environ)
        if v != "-":
            for kv in v.split(","):
                k, val = kv.split("=", 1)
                env[k] = val
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup", str(a.warmup), "--no-cpu-baseline", "--no-drift",
               "--no-tolerance-path", "--no-kernel-timing"] + a.args.split()
        out = subprocess.run(cmd, env=env, capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if l.startswith("{")]
        ms = json.loads(line[-1])["ms_per_step"] if line else float("nan")
        res[v].append(ms)
        print(f"round {r} {v}: {ms} ms", flush=True)
for v, ms in res.items():
    print(f"{v:50s} median {statistics.median(ms):8.2f} ms   all {ms}")
