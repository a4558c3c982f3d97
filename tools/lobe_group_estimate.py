"""How many sources would a GROUP-level reach test spare the per-lane tests, if the 64 sources of a batch were sub-sorted by
heading into groups of G?  CPU model on the headline population (positions and headings of bench.synthetic_population)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synthetic_population
from oracle import csf_oracle as orc

n, box = 16384, 200.0
s0, off, dq = synthetic_population(n, box)
x, y, psi = s0[:, 0], s0[:, 1], s0[:, 2]
P = orc.default_params("twod")
T = np.log(n / 2.0 ** -24)
hf = P.hfov / 2
def reach(cphi, s2):
    e = P.e_0 - P.e_1 * s2; sa = P.sigma_0 + P.sigma_1 * s2; sb = P.sigma_2 + P.sigma_3 * s2
    return T * (sa - sb / 2 + sb / 2 * cphi) / np.sqrt(1 - (e * cphi) ** 2)
phis = np.linspace(0, np.pi, 181)
Rmax_phi = np.max([reach(np.cos(phis), s2) for s2 in np.linspace(0, 1, 21)], axis=0)   # bound over the relative heading
print("T", T, "reach ahead/side/behind (max over s2):", Rmax_phi[0], Rmax_phi[90], Rmax_phi[180], "far radius", Rmax_phi.max())
# spatial order: cells of 0.5 m along a Hilbert curve is what the engine does; a Morton order is close enough for a count
def morton(ix, iy):
    k = np.zeros(ix.shape, dtype=np.int64)
    for b in range(12):
        k |= ((ix >> b) & 1) << (2 * b) | ((iy >> b) & 1) << (2 * b + 1)
    return k
order = np.argsort(morton((x * 2).astype(np.int64) + 64, (y * 2).astype(np.int64) + 64), kind="stable")
rng = np.random.default_rng(0)
recv = rng.choice(n, 256, replace=False)
for G, Q in ((64, 1), (64, 4), (64, 8), (16, 1), (8, 1)):
    tested = kept = cand = lanes = lanes0 = 0
    quad = np.floor(np.mod(psi, 2 * np.pi) / (2 * np.pi / Q)).astype(np.int64)
    order_q = order[np.argsort(quad[order], kind="stable")] if Q > 1 else order      # heading sector first, then the spatial order
    for b in range(n // 64):
        idx = order_q[b * 64:(b + 1) * 64]
        idx = idx[np.argsort(psi[idx])]                       # sub-sort by heading
        for g in range(64 // G):
            sub = idx[g * G:(g + 1) * G]
            cx, cy = x[sub].mean(), y[sub].mean()
            r = np.hypot(x[sub] - cx, y[sub] - cy).max()
            # heading interval of the group (circular: the smallest arc that holds them)
            a = np.sort(np.mod(psi[sub], 2 * np.pi)); gaps = np.diff(np.r_[a, a[0] + 2 * np.pi]); k = gaps.argmax()
            arc = 2 * np.pi - gaps[k]; mid = np.mod(a[(k + 1) % G] + arc / 2, 2 * np.pi)
            for j in recv:
                dx, dy = x[j] - cx, y[j] - cy
                D = np.hypot(dx, dy)
                if D <= r + 1e-9:
                    skip = False
                else:
                    # bearing of the receiver in a source's frame: theta - psi_i, within +- (arc / 2 + asin(r / D)) of theta - mid
                    th = np.arctan2(dy, dx)
                    c0 = np.abs(np.mod(th - mid + np.pi, 2 * np.pi) - np.pi)
                    half = arc / 2 + np.arcsin(min(1.0, r / D))
                    lo, hi = max(0.0, c0 - half), min(np.pi, c0 + half)
                    i0, i1 = int(np.floor(lo / np.pi * 180)), int(np.ceil(hi / np.pi * 180))
                    skip = D - r > Rmax_phi[i0:i1 + 1].max()
                # as the engine counts: every lane of a group that is neither beyond reach nor wholly outside the field of view
                if D > r + 1e-9:
                    bc = np.abs(np.mod(np.arctan2(cy - y[j], cx - x[j]) - psi[j] + np.pi, 2 * np.pi) - np.pi)
                    outside = bc - np.arcsin(min(1.0, r / D)) > hf
                else:
                    outside = False
                lanes += 0 if (skip or outside) else G
                lanes0 += 0 if (outside or (D > r and D - r > Rmax_phi.max())) else G
                ang = np.arctan2(y[sub] - y[j], x[sub] - x[j]) - psi[j]
                infov = np.abs(np.mod(ang + np.pi, 2 * np.pi) - np.pi) <= hf
                infov &= sub != j
                rho = np.hypot(x[sub] - x[j], y[sub] - y[j])
                cphi = ((x[j] - x[sub]) * np.cos(psi[sub]) + (y[j] - y[sub]) * np.sin(psi[sub])) / np.maximum(rho, 1e-9)
                s2 = np.sin(psi[sub] - psi[j]) ** 2
                k_ = infov & (rho <= reach(cphi, s2))
                cand += infov.sum()
                if not skip:
                    tested += infov.sum()
                assert not (skip and k_.any())
                kept += k_.sum()
    print(f"heading sectors {Q}, groups of {G}: sources in the field of view per receiver {cand / recv.size:.0f}, still tested {tested / recv.size:.0f} ({tested / cand:.2f}), kept {kept / recv.size:.0f}; lanes through the tests {lanes / recv.size:.0f} (isotropic far radius only: {lanes0 / recv.size:.0f})")
