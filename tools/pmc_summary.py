#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output directories: last dispatch of the kernel matching a substring."""
import csv, glob, sys
pat = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
        if not rows: continue
        last = max(int(r["Dispatch_Id"]) for r in rows)
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                print(f"{r['Counter_Name']:28s} {float(r['Counter_Value']):.4e}")
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
        if rows:
            r = rows[-1]
            print(f"  duration_us {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f}  ({d})")
