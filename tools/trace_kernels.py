#!/usr/bin/env python
"""Fold a rocprofv3 kernel_trace.csv into one line per distinct (kernel, grid, workgroup) in first-appearance order: launches, average
duration, LDS, registers.  Used to read which Tensile kernels (macro tile, depthU, wave grid -- all in the kernel name) the vendor
library picks for the pipeline's shapes next to this repo's kernels (tools/vs_library_gemm.py).  A measurement tool."""
import csv, sys
from collections import OrderedDict

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2].split(",") if len(sys.argv) > 2 else None
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
acc = OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if pat and not any(p in n for p in pat):
        continue
    key = (n, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = acc.setdefault(key, dict(n=0, t=0.0, mn=1e30, lds=r.get("LDS_Block_Size", "?"), vgpr=r.get("VGPR_Count", "?"), agpr=r.get("Accum_VGPR_Count", "?"), sgpr=r.get("SGPR_Count", "?")))
    a["n"] += 1; a["t"] += d; a["mn"] = min(a["mn"], d)
for (n, g, w), a in acc.items():
    print(f"{a['n']:6d} x avg {a['t'] / a['n']:8.1f} us min {a['mn']:8.1f}  grid {g:>8} wg {w:>4} lds {a['lds']:>6} vgpr {a['vgpr']:>3} agpr {a['agpr']:>3} sgpr {a['sgpr']:>3}  {n[:400]}")
