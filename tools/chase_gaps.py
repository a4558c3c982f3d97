#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of a run with side-by-side ticks: per tick, microseconds between the pair launch's end and the per-agent
kernel's end, between that and the next pair launch's start, pair start to pair start.   tools/chase_gaps.py TRACE.csv"""
import csv
import statistics
import sys

rows = []
with open(sys.argv[1], newline="") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
pairs = [r for r in rows if "pair_cull_kernel" in r[2]]
agents = [r for r in rows if "agent_chase_kernel" in r[2] or "agent_kernel" in r[2]]
gates = [r for r in rows if "chase_gate" in r[2]]
print("pair launches", len(pairs), "per-agent launches", len(agents), "gates", len(gates), "queues", sorted({r[3] for r in pairs}))
a_i = 0
tail, gap, period, alen, astart = [], [], [], [], []
for k in range(len(pairs) - 1):
    p, q = pairs[k], pairs[k + 1]
    ag = [a for a in agents if p[0] <= a[0] <= q[0] + 1]
    if len(ag) != 1:
        continue
    a = ag[0]
    tail.append((a[1] - p[1]) / 1e3)
    gap.append((q[0] - max(a[1], p[1])) / 1e3)
    period.append((q[0] - p[0]) / 1e3)
    alen.append((a[1] - a[0]) / 1e3)
    astart.append((a[0] - p[1]) / 1e3)
med = statistics.median
print(f"ticks {len(tail)}: pair start -> next pair start {med(period):.1f} us; per-agent kernel starts {med(astart):+.1f} us relative to the pair launch's end, "
      f"lasts {med(alen):.1f}, ends {med(tail):+.1f} after it; then {med(gap):.1f} us until the next pair launch starts")
