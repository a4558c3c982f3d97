#!/usr/bin/env python3
"""Would classifying the sources against the field-of-view cone in groups of 32 or 16 instead of 64 save per-lane tests?  (CPU only.)

The cull-first pair kernel (csf_pair.hip) classifies every batch of 64 binned source records from its bounding circle - outside
the receiver's field-of-view cone or beyond the far-field radius: skipped; else: every source of it goes through the per-lane
test (keep_x2) - and puts 5.5 sources through that test for every pair it evaluates (profiles/r4_v10_pair_kernel_pmc.json:
93.6 M tested, 16.9 M evaluated per launch).  This script bins the headline population (16 384 in 200 m x 200 m, Hilbert order
of 0.5 m cells, as csf_bin.hip does) and counts, for 512 random receivers, the sources in groups that are NOT skipped, for
groups of 64, 32 and 16."""
import sys

import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
n, box = 16384, 200.0
s0, off, dq = bench.synthetic_population(n, box, seed=0)
x, y, psi = s0[:,0], s0[:,1], s0[:,2]
# Hilbert order of 0.5 m cells
def hilbert(ix, iy, order=10):
    d = np.zeros(ix.shape, dtype=np.int64)
    x = ix.copy(); y = iy.copy()
    s = 1 << (order - 1)
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64); ry = ((y & s) > 0).astype(np.int64)
        d += s * s * ((3 * rx) ^ ry)
        # rotate
        m = ry == 0
        flip = m & (rx == 1)
        x = np.where(flip, s - 1 - x, x); y = np.where(flip, s - 1 - y, y)
        x2 = np.where(m, y, x); y2 = np.where(m, x, y)
        x, y = x2, y2
        s >>= 1
    return d
ix = np.floor(x / 0.5).astype(np.int64); iy = np.floor(y / 0.5).astype(np.int64)
perm = np.argsort(hilbert(ix, iy), kind='stable')
xs, ys = x[perm], y[perm]
rfar = 154.7
ch = np.cos(np.pi / 3)
def circles(B):
    xb = xs.reshape(-1, B); yb = ys.reshape(-1, B)
    x0, x1, y0, y1 = xb.min(1), xb.max(1), yb.min(1), yb.max(1)
    cx, cy = (x0 + x1) / 2, (y0 + y1) / 2
    r = 0.5 * np.hypot(x1 - x0, y1 - y0) + 0.07
    return cx, cy, r
rng = np.random.default_rng(1)
recv = rng.choice(n, 512, replace=False)
for B in (64, 32, 16):
    cx, cy, r = circles(B)
    tested = 0; kept_in = 0
    for j in recv:
        ex, ey = cx - x[j], cy - y[j]
        D = np.hypot(ex, ey)
        far = D > rfar + r
        sa = np.minimum(r / np.maximum(D, 1e-9), 1.0); ca = np.sqrt(1 - sa * sa)
        cb = (ex * np.cos(psi[j]) + ey * np.sin(psi[j])) / np.maximum(D, 1e-9)
        sb = np.abs(np.cos(psi[j]) * ey - np.sin(psi[j]) * ex) / np.maximum(D, 1e-9)
        apart = D > r + 0.05
        out = far | (apart & (cb < ca) & ((cb * ca + sb * sa) < ch - 1e-4))
        tested += B * int((~out).sum())
    # kept: tracked & within reach (approx: within FOV and within rfar... reach test keeps fewer; use evaluated count from device: ~1032 per receiver)
    print(B, "tested per receiver", tested / len(recv))
