#!/usr/bin/env python
"""Ablation for the attention kernel's output write amplification (profiles: WRITE_SIZE 1.66x the output bytes at d = 40: the
80-byte head segments of a 640-byte row come from 8 workgroups).  A debug build of the same kernels that stores O compactly as
[B, H, Nq, D] -- every workgroup writes one contiguous, line-aligned 10 KB block: no partial lines at all -- is timed against
the product library, alternating child processes on one box.  The debug library (tools/dbg/, built from a patched copy of
csrc/attention.hip, not shipped) produces the wrong LAYOUT by construction; timing only.  Usage: ab_attn_store.py [rounds]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dbg = os.path.join(ROOT, "tools", "dbg", "libgmd_attn_compact_o.so")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
res = {"prod": {}, "compact": {}}
for r in range(rounds):
    for name, lib in (("prod", None), ("compact", dbg)):
        env = dict(os.environ)
        if lib:
            env["GMD_LIB_OVERRIDE"] = lib
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_attn.py")], env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            m = re.match(r"attn (.*): +([0-9.]+) us", line)
            if m:
                res[name].setdefault(m.group(1), []).append(float(m.group(2)))
for shape in res["prod"]:
    a, b = min(res["prod"][shape]), min(res["compact"][shape])
    print(f"{shape:40s} product layout {a:8.1f} us   compact output {b:8.1f} us   ({100 * (b - a) / a:+.1f} %)")
