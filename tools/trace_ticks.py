#!/usr/bin/env python3
"""What runs between two pair launches?  From a rocprofv3 kernel trace (*_kernel_trace.csv): the ticks (pair launch to pair launch) are
grouped by their SEQUENCE of kernels; for the commonest sequences, per kernel the median start relative to the pair launch's start and
the median duration, then the median gap to the next pair launch and the tick.  Copies the runtime enqueues (memcpy / fill kernels) show
up as kernels too.      tools/trace_ticks.py TRACE.csv [pair-kernel-substring] [--from K] [--top 3] [--list N]
--list N: the last N ticks one by one instead - the period and every kernel's start (relative to the pair launch's start) and duration."""
import collections
import csv
import statistics
import sys

args = sys.argv[1:]
top = int(args[args.index("--top") + 1]) if "--top" in args else 3
first = int(args[args.index("--from") + 1]) if "--from" in args else 0
listn = int(args[args.index("--list") + 1]) if "--list" in args else 0
pos = [a for k, a in enumerate(args) if not a.startswith("--") and (k == 0 or args[k - 1] not in ("--top", "--from", "--list"))]
key = pos[1] if len(pos) > 1 else "pair_cull_kernel"
rows = []
with open(pos[0], newline="") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
starts = [k for k, r in enumerate(rows) if key in r[2]][first:]


def short(name):
    name = name.split("(")[0]
    for cut in ("void ", "csf::"):
        name = name.replace(cut, "")
    return name[:58]


if listn:
    for a, b in list(zip(starts[:-1], starts[1:]))[-listn:]:
        t0 = rows[a][0]
        print(f"{(rows[b][0] - t0) / 1e3:8.1f} us  " + "  ".join(f"{short(rows[k][2])[:24]}@{(rows[k][0] - t0) / 1e3:.1f}+{(rows[k][1] - rows[k][0]) / 1e3:.1f}[q{rows[k][3]}]" for k in range(a, b)))
    sys.exit(0)
ticks = collections.defaultdict(list)
for a, b in zip(starts[:-1], starts[1:]):
    seq = tuple(short(rows[k][2]) for k in range(a, b))
    ticks[seq].append((a, b))
med = statistics.median
print(f"{len(starts)} launches of *{key}*; {len(ticks)} distinct kernel sequences between two of them")
for seq, lst in sorted(ticks.items(), key=lambda kv: -len(kv[1]))[:top]:
    period = med((rows[b][0] - rows[a][0]) / 1e3 for a, b in lst)
    print(f"\n{len(lst)} ticks, pair start -> next pair start {period:.1f} us (median):")
    prev_end = None
    for j, name in enumerate(seq):
        st = med((rows[a + j][0] - rows[a][0]) / 1e3 for a, b in lst)
        du = med((rows[a + j][1] - rows[a + j][0]) / 1e3 for a, b in lst)
        gap = "" if j == 0 else f"  ({med((rows[a + j][0] - max(rows[a + i][1] for i in range(j))) / 1e3 for a, b in lst):+.1f} after what ran before)"
        q = {rows[a + j][3] for a, b in lst}
        print(f"  {st:8.1f}  {du:7.1f} us  {name}  [queue {','.join(sorted(q))}]{gap}")
    tail = med((rows[b][0] - max(rows[k][1] for k in range(a, b))) / 1e3 for a, b in lst)
    print(f"  then {tail:.1f} us until the next pair launch starts")
