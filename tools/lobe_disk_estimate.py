"""The cheap form of tools/lobe_group_estimate.py: the reach lobe inside an offset disk, one circle test per group (CPU model)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_population
from oracle import csf_oracle as orc
n, box = 16384, 200.0
s0, off, dq = synthetic_population(n, box)
x, y, psi = s0[:, 0], s0[:, 1], s0[:, 2]
P = orc.default_params("twod")
T = np.log(n / 2.0 ** -24); hf = P.hfov / 2
def reach(cphi, s2):
    e = P.e_0 - P.e_1 * s2; sa = P.sigma_0 + P.sigma_1 * s2; sb = P.sigma_2 + P.sigma_3 * s2
    return T * (sa - sb / 2 + sb / 2 * cphi) / np.sqrt(1 - (e * cphi) ** 2)
phis = np.linspace(0, np.pi, 721)
R = np.max([reach(np.cos(phis), s2) for s2 in np.linspace(0, 1, 41)], axis=0)
# smallest disk (centre on the heading axis at offset d) that holds the lobe
best = None
for d in np.linspace(0, 80, 321):
    rad = np.sqrt((R * np.cos(phis) - d) ** 2 + (R * np.sin(phis)) ** 2).max()
    if best is None or rad < best[1]: best = (d, rad)
d0, rad0 = best
print("lobe within a disk of radius %.1f m centred %.1f m ahead of the source (far radius %.1f): area ratio %.2f" % (rad0, d0, R.max(), rad0 ** 2 / R.max() ** 2))
def morton(ix, iy):
    k = np.zeros(ix.shape, dtype=np.int64)
    for b in range(12):
        k |= ((ix >> b) & 1) << (2 * b) | ((iy >> b) & 1) << (2 * b + 1)
    return k
order = np.argsort(morton((x * 2).astype(np.int64) + 64, (y * 2).astype(np.int64) + 64), kind="stable")
rng = np.random.default_rng(0)
recv = rng.choice(n, 256, replace=False)
for G in (64, 16, 8):
    lanes = lanes0 = 0
    for b in range(n // 64):
        idx = order[b * 64:(b + 1) * 64]
        idx = idx[np.argsort(psi[idx])]
        for g in range(64 // G):
            sub = idx[g * G:(g + 1) * G]
            cx, cy = x[sub].mean(), y[sub].mean()
            r = np.hypot(x[sub] - cx, y[sub] - cy).max()
            qx, qy = x[sub] + d0 * np.cos(psi[sub]), y[sub] + d0 * np.sin(psi[sub])     # centres of the members' reach disks
            mx, my = qx.mean(), qy.mean()
            rho = rad0 + np.hypot(qx - mx, qy - my).max()
            D = np.hypot(x[recv] - cx, y[recv] - cy)
            bc = np.abs(np.mod(np.arctan2(cy - y[recv], cx - x[recv]) - psi[recv] + np.pi, 2 * np.pi) - np.pi)
            outside = (D > r) & (bc - np.arcsin(np.minimum(1.0, r / np.maximum(D, 1e-9))) > hf)
            far = (D > r) & (D - r > R.max())
            disk = np.hypot(x[recv] - mx, y[recv] - my) > rho
            lanes0 += G * (~(outside | far)).sum()
            lanes += G * (~(outside | far | disk)).sum()
    print(f"groups of {G}: lanes through the tests per receiver {lanes / recv.size:.0f} (far radius only {lanes0 / recv.size:.0f})")
