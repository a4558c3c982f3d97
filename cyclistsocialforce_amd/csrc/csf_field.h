// csf_field.h — the repulsive force fields and the field-of-view test as device functions (fp32, trig-free): TwoDBicycle
// field (vehicle.py:1560-1648) one pair per lane and two pairs per lane on packed arithmetic, the reach test, the Bicycle
// field (vehicle.py:1054-1147).  Shared by the pair kernels (csf_pair.hip) and the one-launch tick of small populations
// (csf_tick.hip).
#pragma once
#include "csf_dev.h"

namespace csf {

typedef float v2f __attribute__((ext_vector_type(2)));

struct Recv {
    float x, y, c, s;
};

__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ v2f rsq2(v2f x) { return v2f{fast_rsq(x.x), fast_rsq(x.y)}; }
__device__ __forceinline__ v2f fabs2(v2f x) { return __builtin_elementwise_max(x, -x); }  // one v_pk_max_f32

// intersection.py:690-745 for receiver r and source (dx, dy) = receiver - source.
// The receiver ignores the source when the bearing of the source, relative to the receiver's heading, is
// outside +-hfov/2 (hfov of the SOURCE's class, :733-735; one class per engine), when it is to the left under
// priority-to-the-right, or when it is the receiver itself / coincident (rho = 0).
// With t = rho cos(bearing) the test |bearing| <= hfov/2 is  t|t| - cos^2(hfov/2) rho^2 >= 0  for hfov <= pi and
// t|t| + cos^2(hfov/2) rho^2 >= 0  beyond; chs carries the sign.  The comparison is strict so that rho = 0 fails.
template <bool P2R>
__device__ __forceinline__ bool tracked(float chs, const Recv &r, float dx, float dy, float r2) {
    float t = -(dx * r.c + dy * r.s);
    bool in = (t * fabsf(t) + chs * r2) > 0.0f;
    if (P2R) in = in & !((r.s * dx - r.c * dy) > 0.0f);  // rho * sin(bearing) > 0: the source is to the left
    return in;
}

// vehicle.py:1560-1648: force of source (record q) on receiver r, returned as magnitude F and an
// unnormalised direction (gx, gy) with F already holding 1/|g|.  (dx, dy) = receiver - source.
__device__ __forceinline__ void field_twod(const PairConsts &k, const Recv &r, const float4 q, float dx,
                                           float dy, float r2, float &F, float &gx, float &gy) {
    float inv = fast_rsq(r2), rho = r2 * inv;
    float srel = q.w * r.c - q.z * r.s;               // sin(psi0 - psi)            :1595
    float s2 = srel * srel;
    float sga = k.sg0 + k.sg1 * s2;                   // :1604-1606
    float sgb = k.sg2 + k.sg3 * s2;                   // :1607-1609
    float e = k.e0 - k.e1 * s2;                       // :1612
    float cphi = (dx * q.z + dy * q.w) * inv;         // cos(phi1 - psi0)          :1618-1620
    float sphi = (dy * q.z - dx * q.w) * inv;         // sin(phi1 - psi0)
    // half-angle roots without cancellation: big = sqrt((1+|c|)/2), small = |s| / (2 big)
    float a = 0.5f + 0.5f * fabsf(cphi);
    float rs = fast_rsq(a);
    float big = a * rs, small = 0.5f * fabsf(sphi) * rs;
    float sg = __builtin_amdgcn_fmed3f(sphi * 1e38f, -1.0f, 1.0f);  // np.sign(phi), 0 at phi = 0 :1625
    bool pos = cphi >= 0.0f;
    float bs = big * sg, al = 0.5f * sphi * rs;
    float h1 = pos ? small : big;                     // sqrt((1 - cos phi)/2)       :1624
    float h2s = pos ? bs : al;                        // sqrt((1 + cos phi)/2) * sign(phi)
    float sigma = sga - sgb * h1;                     // :1624
    float dsig = -0.5f * sgb * h2s;                   // :1625
    float ec = e * cphi;
    float q2 = 1.0f - ec * ec;
    // :1631-1642 with the positive factor P / (sigma^2 q) taken out of both polar components and the
    // rotation by phi1 written with rho*cos(phi1) = dx, rho*sin(phi1) = dy
    float grho = q2 * sigma;
    float gphi = e * ec * sphi * sigma - q2 * dsig;
    gx = grho * dx - gphi * dy;
    gy = grho * dy + gphi * dx;
    float ig = fast_rsq(gx * gx + gy * gy);
    float qos = q2 * fast_rsq(grho * sigma);          // q / sigma = q^2 / sqrt(q^2 sigma^2): one rsq, no sqrt + rcp
    float P = fast_exp2(k.lf0 - k.kexp * (rho * qos));  // f_0 exp(-rho q / sigma)            :1628
    F = P * ig;                                       // :1644-1646: |F| = P
}

// The same field for TWO sources per lane (components .x / .y), float2 arithmetic -> v_pk_* instructions.
// FULL: both sources of every lane are real, tracked pairs (rho > 0); otherwise valid0 / valid1 mask the lanes
// of a partial batch, whose dummy record may coincide with the receiver.  Accumulates into (ax, ay).
// NEARFLAG: near0 / near1 report the pairs closer than k.rnear (pair_cull_kernel corrects them from the precise records).
template <bool FULL, bool NEARFLAG = false, bool LO = false>
__device__ __forceinline__ void field_twod_x2(const PairConsts &k, const Recv &r, const v2f qx, const v2f qy,
                                              const v2f qc, const v2f qs, bool valid0, bool valid1, float &ax,
                                              float &ay, bool *near0 = nullptr, bool *near1 = nullptr,
                                              const v2f lx = v2f{0.f, 0.f}, const v2f ly = v2f{0.f, 0.f}) {
    // (lx, ly): an optional correction of (dx, dy), the difference of the low parts of two-float positions (csf_tick.hip)
    v2f dx = r.x - qx, dy = r.y - qy;                 // vehicle.py:1615-1616
    if (LO) dx = dx + lx, dy = dy + ly;
    v2f r2 = dx * dx + dy * dy;
    if (NEARFLAG) {
        *near0 = valid0 & (r2.x < k.rnear2);
        *near1 = valid1 & (r2.y < k.rnear2);
    }
    if (!FULL) r2 = v2f{fmaxf(r2.x, 1e-30f), fmaxf(r2.y, 1e-30f)};
    v2f inv = rsq2(r2), rho = r2 * inv;
    v2f srel = qs * r.c - qc * r.s;                   // sin(psi0 - psi)            :1595
    v2f s2 = srel * srel;
    v2f sga = k.sg0 + k.sg1 * s2, sgb = k.sg2 + k.sg3 * s2, e = k.e0 - k.e1 * s2;   // :1604-1612
    v2f cphi = (dx * qc + dy * qs) * inv, sphi = (dy * qc - dx * qs) * inv;           // :1618-1621
    v2f a = 0.5f + 0.5f * fabs2(cphi);
    v2f rs = rsq2(a);
    v2f big = a * rs, hrs = 0.5f * rs;
    v2f al = sphi * hrs;
    const v2f blown = sphi * 1e38f;                   // one packed multiply, then clamp to -1, 0, +1
    v2f sg{__builtin_amdgcn_fmed3f(blown.x, -1.0f, 1.0f), __builtin_amdgcn_fmed3f(blown.y, -1.0f, 1.0f)};
    v2f small = al * sg, bs = big * sg;                // |sphi| hrs and sign(phi) big          :1624-1625
    const bool p0 = cphi.x >= 0.0f, p1 = cphi.y >= 0.0f;
    v2f h1{p0 ? small.x : big.x, p1 ? small.y : big.y};
    v2f h2s{p0 ? bs.x : al.x, p1 ? bs.y : al.y};
    v2f sigma = sga - sgb * h1;
    v2f hd = 0.5f * sgb * h2s;                        // = -dsig
    v2f ec = e * cphi;
    v2f q2 = 1.0f - ec * ec;
    v2f grho = q2 * sigma;                            // :1631-1642, common factor P/(sigma^2 q) removed
    v2f gphi = (e * ec) * (sphi * sigma) + q2 * hd;
    v2f gx = grho * dx - gphi * dy, gy = grho * dy + gphi * dx;
    v2f ig = rsq2(gx * gx + gy * gy);
    v2f qos = q2 * rsq2(grho * sigma);                // q / sigma = q^2 / sqrt(q^2 sigma^2): one rsq, no sqrt + rcp
    v2f ex = k.lf0 - k.kexp * (rho * qos);            // :1628
    v2f F = v2f{fast_exp2(ex.x), fast_exp2(ex.y)} * ig;   // :1644-1646
    if (!FULL) F = v2f{valid0 ? F.x : 0.0f, valid1 ? F.y : 0.0f};
    const v2f cx = F * gx, cy = F * gy;
    ax += cx.x + cx.y;
    ay += cy.x + cy.y;
}

// Reach test for TWO sources per lane (csf_engine.hip: update_far_radius): keep a pair unless its contribution is provably
// below far_eps f_0 / n,
//     rho^2 - e^2 X^2 > T^2 (sigma_a - sigma_b / 2 + (sigma_b / 2) X / rho)^2,   X = rho cos(phi) = (dx, dy) . (cos psi0, sin psi0),
// which needs the three pair scalars (rho^2, X, s2) and one rsq: ~40 % of the cost of the field (vehicle.py:1604-1628),
// and removes three of four pairs inside the field of view.  FOV adds the exact mask of intersection.py:690-745 for
// batches that are not wholly inside the field of view.
template <bool FOV, bool P2R>
__device__ __forceinline__ void keep_x2(const PairConsts &k, const Recv &r, const v2f qx, const v2f qy, const v2f qc,
                                        const v2f qs, bool &k0, bool &k1) {
    const v2f dx = r.x - qx, dy = r.y - qy;
    const v2f r2 = dx * dx + dy * dy;
    const v2f X = dx * qc + dy * qs;
    const v2f srel = qs * r.c - qc * r.s;
    const v2f s2 = srel * srel;
    const v2f e = k.e0 - k.e1 * s2, A = k.tA0 + k.tA1 * s2, B = k.tB0 + k.tB1 * s2;
    const v2f c = X * rsq2(r2);                        // r2 = 0 (the receiver itself): NaN -> kept here, masked by FOV
    const v2f S = A + B * c;
    const v2f eX = e * X;
    const v2f f = (r2 - eX * eX) - S * S;
    k0 = !(f.x > 0.0f);
    k1 = !(f.y > 0.0f);
    if (FOV) {
        const v2f t = -(dx * r.c + dy * r.s);
        const v2f g = t * fabs2(t) + k.chs * r2;
        k0 = k0 & (g.x > 0.0f);
        k1 = k1 & (g.y > 0.0f);
        if (P2R) {
            const v2f side = r.s * dx - r.c * dy;      // rho sin(bearing) > 0: the source is to the left
            k0 = k0 & !(side.x > 0.0f);
            k1 = k1 & !(side.y > 0.0f);
        }
    }
}

// vehicle.py:1054-1147: older elliptic field of base Bicycle; q2v = (e, 1/sqrt(1-e^2)) of the source.
__device__ __forceinline__ void field_bicycle(const PairConsts &k, const float4 q, const float2 q2v, float dx,
                                              float dy, float r2, float &F, float &gx, float &gy) {
    float inv = fast_rsq(r2), rho = r2 * inv;
    float c0 = (dx * q.z + dy * q.w) * inv;           // cos(phi - psi0)            :1129
    float s0 = (dy * q.z - dx * q.w) * inv;
    float w = (1.0f - q2v.x * c0) * q2v.y;            // (1 - e cos)/sqrt(1-e^2)
    float P = fast_exp2(k.lf0 - k.kexp * (rho * w * k.ipd));  // (p0/p_decay) exp(-b)  :1095-1101, :1132
    float frho = w, fphi = q2v.x * s0 * q2v.y;        // :1135-1140 (common factor P)
    gx = (frho * dx - fphi * dy) * inv;               // :1144-1145
    gy = (frho * dy + fphi * dx) * inv;
    F = P;
}

}  // namespace csf
