#!/usr/bin/env python3
"""Which launches are the slow ones?  From a rocprofv3 kernel trace (*_kernel_trace.csv): per kernel the launch count, min / median /
mean / max duration, and the five longest launches with their call index (0 = the kernel's first launch in the process), their
position in the process's launch sequence and what ran right before them.  A `kernel_stats.csv` mean hides a first launch that
loads its code object or one that follows a re-binning's read-back; this names them.
    tools/kernel_outliers.py TRACE.csv"""
import csv
import statistics
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1], newline="") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
by = defaultdict(list)
short = lambda s: s.split("(")[0][:110]   # noqa: E731
for seq, (t0, t1, name) in enumerate(rows):
    prev = short(rows[seq - 1][2]) if seq else "-"
    gap = (t0 - rows[seq - 1][1]) / 1e3 if seq else 0.0
    by[name].append((t1 - t0, len(by[name]), seq, prev, gap))
for name, v in sorted(by.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    d = [x[0] / 1e3 for x in v]
    print(f"{short(name)}\n  launches {len(d)}  min {min(d):.2f}  median {statistics.median(d):.2f}  mean {statistics.fmean(d):.2f}  max {max(d):.2f} us")
    for dur, idx, seq, prev, gap in sorted(v, reverse=True)[:5]:
        print(f"    {dur / 1e3:9.2f} us  call #{idx} of this kernel, launch #{seq} of the process, {gap:.1f} us after {prev}")
