#!/usr/bin/env python
"""Fold a rocprofv3 --kernel-trace CSV by (kernel, grid size, workgroup size): launches, average / total device time.  The grid
size tells the shapes of one kernel apart (e.g. the level-0 feed-forward launches among all ring-GEMM launches), so in-situ
durations inside the replayed graphs can be compared between variants.  Usage: rocprof_shapes.py <kernel_trace.csv> [min_total_ms]"""
import collections, csv, re, sys

agg = collections.defaultdict(lambda: [0, 0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*\)$", "", name)
        key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))
        a = agg[key]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
lim = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(v[1] for v in agg.values())
print(f"{'kernel':70s} {'grid':>9s} {'wg':>5s} {'launches':>8s} {'avg_us':>8s} {'total_ms':>9s} {'share':>6s}")
for (name, grid, wg), (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if ns / 1e6 >= lim:
        print(f"{name[:70]:70s} {grid:>9s} {wg:>5s} {n:8d} {ns / n / 1e3:8.2f} {ns / 1e6:9.2f} {ns / tot:6.3f}")
