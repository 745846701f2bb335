#!/usr/bin/env python
"""Samples the GPU's power, shader clock and utilisation (rocm-smi, no HIP context in this process) while a command runs, and
prints a summary per phase of busy / idle.  Usage: power_trace.py [--period 0.25] -- <command ...>
The question it answers (DESIGN.md §4.1 "Clock"): is the sustained denoising loop power / clock limited?"""
import json, subprocess, sys, time

period = 0.25
argv = sys.argv[1:]
if argv and argv[0] == "--period":
    period = float(argv[1]); argv = argv[2:]
assert argv and argv[0] == "--", __doc__
cmd = argv[1:]


def sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showuse", "--showtemp", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        c = d[sorted(k for k in d if k.startswith("card"))[0]]
    except Exception as e:  # noqa
        return {"error": repr(e)}
    g = lambda *names: next((c[k] for k in c for n in names if n.lower() in k.lower()), None)
    return {"power": g("Average Graphics Package Power", "Current Socket Graphics Package Power", "Power (W)"), "sclk": g("sclk clock speed"),
            "mclk": g("mclk clock speed"), "use": g("GPU use (%)"), "temp": g("Temperature (Sensor junction)", "Temperature (Sensor edge)"), "raw_keys": None}


first = sample()
print("first sample:", first, flush=True)
p = subprocess.Popen(cmd)
rows, t0 = [], time.time()
while p.poll() is None:
    s = sample(); s["t"] = round(time.time() - t0, 2); rows.append(s)
    time.sleep(period)
print("exit code", p.returncode)


def num(v):
    try:
        return float(str(v).strip("()MHzmhzWwcC% "))
    except Exception:
        return None


busy = [r for r in rows if (num(r.get("use")) or 0) >= 90]
for name, rs in (("all samples", rows), ("samples with GPU use >= 90 %", busy)):
    for key in ("power", "sclk", "mclk", "temp"):
        vs = sorted(v for v in (num(r.get(key)) for r in rs) if v is not None)
        if vs:
            print(f"{name:30s} {key:5s} n={len(vs):4d} min {vs[0]:8.1f} median {vs[len(vs) // 2]:8.1f} p90 {vs[int(len(vs) * 0.9)]:8.1f} max {vs[-1]:8.1f}")
print("trace (t, use, power, sclk):")
for r in rows[:: max(1, len(rows) // 60)]:
    print(f"  {r.get('t'):7.2f} {str(r.get('use')):>5s} {str(r.get('power')):>8s} {str(r.get('sclk')):>10s}")
