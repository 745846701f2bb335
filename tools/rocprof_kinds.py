#!/usr/bin/env python
"""Fold a rocprofv3 --kernel-trace CSV into the per-kind table bench.py prints (conv3x3 = implicit-GEMM conv kernel plus the
split-K reduction that follows it; gemm_nt likewise; attention), so the rocprof durations can be compared with the
HIP-event durations of bench.py's instrumented step.  Usage: rocprof_kinds.py <kernel_trace.csv>"""
import collections, csv, re, sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
agg = collections.OrderedDict()
def add(kind, ns):
    a = agg.setdefault(kind, [0, 0])
    a[0] += 1; a[1] += ns
last_kind = None
for s, e, name in rows:
    d = e - s
    if "spin_kernel" in name:  # torch.cuda._sleep: the host head start of bench.py's instrumented step, not workload
        continue
    if "splitk_reduce" in name or "gn_slab_kernel" in name:  # (gn_slab: reduction + GroupNorm of gmd_conv3x3_groupnorm, timed
        # by bench.py as part of that conv3x3 call)
        if last_kind:  # belongs to the GEMM / conv launch it completes: add time, not a launch
            agg[last_kind][1] += d
        continue
    # <[element type, ]CONV, ...> for the 16-bit kernels, <CONV, ...> for the float32 ones
    g = re.search(r"gemm_(?:split|f32)_kernel<(true|false)", name) or re.search(r"gemm_(?:ring|bf16|pp|lc)_kernel<[^,<>]+, (true|false)", name)
    if g is None and "conv_patch" in name:  # round 4: conv3x3 with the input patch resident in LDS
        g = re.match(r"(true)", "true")
    # arithmetic family: the default bench command also runs the float32 pipelines (tolerance_path, drift reference), whose
    # launches must not be averaged into the 16-bit kinds
    fam = (" [f32 split]" if ("gemm_split_kernel" in name or "attn_split_kernel" in name) else " [f32 exact]" if "gemm_f32_kernel" in name
           else " [f16]" if "_Float16" in name else "")
    if g and g.group(1) == "true":
        last_kind = "conv3x3" + fam
    elif g or "ff_fused_kernel" in name:
        last_kind = "gemm_nt" + fam
    elif "attn_fwd_kernel" in name or "attn40_kernel" in name or "attn_split_kernel" in name:
        last_kind = "attention" + fam
    else:
        last_kind = None
        m = re.search(r"(?:::)?(\w+)\s*(?:<|\()", name.replace("void ", "").replace("(anonymous namespace)::", ""))
        add("other:" + (m.group(1) if m else name[:40]), d)
        continue
    add(last_kind, d)
tot = sum(v[1] for v in agg.values())
print(f"{'kind':48s} {'launches':>9s} {'total_ms':>10s} {'avg_us':>8s} {'share':>6s}")
for k, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:48s} {n:9d} {ns / 1e6:10.3f} {ns / n / 1e3:8.2f} {ns / tot:6.3f}")
