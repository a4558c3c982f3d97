#!/usr/bin/env python3
"""Read a CSF_CHASE_CLOCK file (tools/chase_clock.py explains the stamps).   tools/chase_clock_read.py FILE"""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
last = int(raw[-1])
ck = raw[:1024].reshape(128, 8).astype(np.float64) / 100.0
rows = []
for r in range(max(last - 60, 1), last - 1):
    a, b = ck[r & 127], ck[(r + 1) & 127]
    if a[1] <= 0 or b[0] > 1e17 or a[0] > 1e17:
        continue
    rows.append([a[1] - a[0], a[3] - a[1], a[4] - a[1], a[5] - a[1], a[6] - a[1], b[0] - a[6], b[0] - a[0]])
rows = np.array([r for r in rows if 0 < r[6] < 1000])
m = np.median(rows, axis=0)
print(f"{sys.argv[1]}: {len(rows)} ticks (of {last} side by side): pair first start -> last arrival {m[0]:.1f}; gate exit {m[1]:+.1f}, per-agent first entry {m[2]:+.1f}, "
      f"last wave past its wait {m[3]:+.1f}, per-agent last end {m[4]:+.1f} (relative to that arrival); next pair start {m[5]:.1f} later; pair start -> pair start {m[6]:.1f}")
