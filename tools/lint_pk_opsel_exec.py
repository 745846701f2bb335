#!/usr/bin/env python
"""Lint of the compiled kernels for the instruction pattern behind the round-4 producer-statistics fault: a packed
float32 VALU instruction whose low result reads a high register (`v_pk_{add,mul,fma}_f32 ... op_sel:[..]`) followed within a few instructions by
a write of EXEC.  Compiles every csrc/*.hip to gfx950 assembly (minutes) and lists the hits per kernel.

    python tools/lint_pk_opsel_exec.py [--window 6] [--jobs 4] [--stamp file] [file.hip ...]
`make -C gm-diffusion_amd/csrc lint` runs it (stamp build/lint.ok, redone when a source changes); __graft_entry__.build() runs that
target, and tests/test_native_abi.py::test_no_swizzled_packed_f32_next_to_an_exec_write keeps it in the CPU suite."""
import argparse, glob, os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# `op_sel:[..]` = the LOW result reads a HIGH register (the form that failed); `op_sel_hi:[1,0]` alone is the broadcast of a low
# register (x - mean in every normalisation kernel), which the bit-exact graph-vs-eager tests have exercised for three rounds
PK = re.compile(r"^\s*v_pk_(add|mul|fma)_f32 .*op_sel:\[")
EXEC_W = re.compile(r"^\s*(s_\w+_saveexec_b64|s_(or|and|andn2|xor|mov|cselect|not)_b64 exec\b|v_cmpx_)")


def lint_file(f, window):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", out, f],
                       check=True, stderr=subprocess.DEVNULL)
        lines = [l.rstrip("\n") for l in open(out)]
    kernel, hits, npk = "?", {}, 0
    ins = []  # (kernel, text) of real instructions
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kernel = m.group(1)
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        ins.append((kernel, t))
    for i, (k, t) in enumerate(ins):
        if PK.match(t):
            npk += 1
            for j in range(i + 1, min(i + 1 + window, len(ins))):
                if ins[j][0] != k:
                    break
                if EXEC_W.match(ins[j][1]):
                    hits.setdefault(k, []).append((t, j - i, ins[j][1]))
                    break
    return f, npk, hits


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--window", type=int, default=6, help="instructions after the packed op in which an EXEC write counts")
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--stamp", default="", help="touch this file when the lint is clean (make lint)")
    ap.add_argument("files", nargs="*")
    a = ap.parse_args()
    files = a.files or sorted(glob.glob(os.path.join(ROOT, "gm-diffusion_amd", "csrc", "*.hip")))
    total = 0
    with ThreadPoolExecutor(max_workers=a.jobs) as ex:
        for f, npk, hits in ex.map(lambda x: lint_file(x, a.window), files):
            n = sum(len(v) for v in hits.values())
            total += n
            print(f"{os.path.basename(f)}: {npk} swizzled packed-f32 instructions, {n} within {a.window} instructions of an EXEC write")
            for k, v in hits.items():
                print(f"  {k[:110]}: {len(v)}")
                for t, d, e in v[:3]:
                    print(f"      {t}   ... +{d}: {e}")
    if total == 0 and a.stamp:
        open(a.stamp, "w").write("clean\n")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
