#!/usr/bin/env python
"""Per-queue occupancy of a rocprofv3 --kernel-trace CSV over the densest window of the run (the timed bench steps): for each HIP
stream / HSA queue the number of dispatches, the union of its kernels' [start, end) intervals (busy time), the idle time between
them and the sum of kernel durations -- tells which stream of the two-stream pipeline is the critical one.
Usage: rocprof_streams.py <kernel_trace.csv> [window_ms]"""
import collections, csv, sys

rows = []
with open(sys.argv[1]) as f:
    rd = csv.DictReader(f)
    qkey = "Stream_Id" if "Stream_Id" in rd.fieldnames else "Queue_Id"
    for r in rd:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[qkey], r["Kernel_Name"]))
rows.sort()
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 1500e6
t_end = rows[-1][1]
t0 = t_end - win
sel = [r for r in rows if r[0] >= t0]
print(f"queue key: {qkey}; window {win / 1e6:.0f} ms ending at the last dispatch; {len(sel)} dispatches")
byq = collections.defaultdict(list)
for s, e, q, n in sel:
    byq[q].append((s, e, n))
allbusy = 0
for q, iv in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    iv.sort()
    busy, cur_s, cur_e, gaps = 0, iv[0][0], iv[0][1], []
    for s, e, n in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = iv[-1][1] - iv[0][0]
    big = sorted(gaps)[-5:] if gaps else []
    print(f"queue {q}: {len(iv):7d} dispatches  span {span / 1e6:8.1f} ms  busy {busy / 1e6:8.1f} ms ({busy / span:5.1%})  sum of durations {sum(e - s for s, e, _ in iv) / 1e6:8.1f} ms  "
          f"gaps: n={len(gaps)} total {sum(gaps) / 1e6:.1f} ms median {sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0:.1f} us largest {[round(g / 1e3) for g in big]} us")
