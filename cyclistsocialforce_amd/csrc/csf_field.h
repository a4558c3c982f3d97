// csf_field.h — the repulsive force fields and the field-of-view test as device functions (fp32, trig-free): TwoDBicycle
// field (vehicle.py:1560-1648) one pair per lane and two pairs per lane on packed arithmetic, the reach test, the Bicycle
// field (vehicle.py:1054-1147); the rare paths that make the field-of-view decision exact.  Used by the pair kernels
// (csf_pair.hip).
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

// The same test with the band of its fp32 rounding (PairConsts::fovA ...), every result ONE compare (csf_pair.hip: ballot1):
// kept - tracked, or within rounding of an edge -, lt - below the upper edge of the band.  A source is MARGINAL when
// kept & lt: it has to be decided as the reference decides it by whoever evaluates it.  rho = 0 is never kept.  The return
// value is the plain fp32 decision.
template <bool P2R>
__device__ __forceinline__ bool tracked_m(const PairConsts &k, float chs, const Recv &r, float dx, float dy, float r2, bool &kept, bool &lt) {
    const float t = -(dx * r.c + dy * r.s);
    const float g = t * fabsf(t) + chs * r2;
    const float band = k.fovA * r2 + k.fovB;
    bool in = g > 0.0f;
    if (!P2R) {
        kept = __builtin_fminf(g + band, r2) > 0.0f;
        lt = g < band;
    } else {
        const float side = r.s * dx - r.c * dy, sb = k.sideA * r2 + k.sideB;
        in = in & !(side > 0.0f);
        kept = (__builtin_fminf(g + band, r2) > 0.0f) & !(side > sb);
        lt = (g < band) | (side > -sb);
    }
    return in;
}

// One source per lane through the very test of keep_x2<FOV> (same arithmetic, so that the variants of the cull-first kernel
// with and without the reach test keep and flag the same sources): kept - cos(bearing) - cos(hfov/2) > -T -, lt - < T.
template <bool P2R>
__device__ __forceinline__ void tracked_c(const PairConsts &k, const Recv &r, float dx, float dy, bool &kept, bool &lt) {
    const float r2 = dx * dx + dy * dy;
    const float inv = fast_rsq(r2);
    const float t = -(dx * r.c + dy * r.s);
    const float T = k.fovT0 + k.fovT1 * inv;
    const float dc = t * inv - k.chk;
    const float z = dc + T;
    kept = z > 0.0f;                                   // (r2 = 0: NaN - the receiver itself is never kept)
    lt = dc < T;
    if (P2R) {
        const float ss = (r.s * dx - r.c * dy) * inv;
        kept = kept & !(ss > T);
        lt = lt | (ss > -T);
    }
}

// intersection.py:690-745 for a pair formed from the PRECISE records (csf_pair.hip: precise_delta).  edge: the pair is within
// rounding of an edge even so (about one in 1e7) - the caller hands it to the per-agent kernel (edge_handover), which
// decides it as the reference does.
template <bool P2R>
__device__ __forceinline__ bool tracked_precise(const PairConsts &k, float chs, const Recv &r, float dx, float dy, float r2, bool &edge) {
    const float t = -(dx * r.c + dy * r.s), rho = fast_sqrt(r2);
    const float g = t * fabsf(t) + chs * r2;
    bool in = g > 0.0f;
    edge = fabsf(g) < k.fovP1 * rho + k.fovP2 * r2;
    if (P2R) {
        const float side = r.s * dx - r.c * dy;
        in = in & !(side > 0.0f);
        edge = edge | (fabsf(side) < k.sideP0 + k.sideP1 * rho);
    }
    edge = edge & (r2 > 0.0f);
    return in;
}

// np.sign(phi) of the TwoD field (vehicle.py:1625) for a pair formed from the precise records: is the receiver within rounding
// of the line AHEAD of the source (behind it the field does not jump)?  rho sin(phi) against the band of tracked_precise's side
// test - the same arithmetic: offsets, and a heading known to 2^-24.
__device__ __forceinline__ bool side_undecided(const PairConsts &k, const float4 q, float dx, float dy, float r2) {
    const float rho = fast_sqrt(r2);
    return (dx * q.z + dy * q.w >= 0.0f) & (fabsf(dy * q.z - dx * q.w) < k.sideP0 + k.sideP1 * rho) & (r2 > 0.0f);
}

// One undecidable pair -> the ring the per-agent kernel reads (csf_dev.h: EdgeRec; csf_agent.hip: COMBINE): the source's
// fp64 position as it is now, its field of view, the pair's force (fx, fy) and whether the caller has added it (seen).
// The source's position: its fp64 state where this device holds it, else (a rank of a sharded run) rebuilt from the record.
// side: np.sign(phi) of the pair is undecided as well - (fx, fy) was evaluated with +1, (fx2, fy2) with -1.
__device__ __forceinline__ void edge_handover(const Dev &d, int32_t a_recv, int32_t a_src, double hfov, float fx, float fy, bool seen,
                                              bool side = false, float fx2 = 0.0f, float fy2 = 0.0f) {
    double xi, yi, psi;
    if (d.state_current) {
        xi = d.src64[a_src];
        yi = d.src64[d.cap + a_src];
        psi = d.src64[2 * d.cap + a_src];
    } else {   // a rank of a sharded run: origin + record + what the record left over = the owner's fp64 position to ~1e-14 m
        const float4 q = d.rec[a_src];
        const float2 o = d.rorg[a_src], lo = d.reclo[a_src];
        xi = (d.ox + (double)o.x) + ((double)q.x + (double)lo.x);
        yi = (d.oy + (double)o.y) + ((double)q.y + (double)lo.y);
        // (a foreign heading is known as the fp32 pair of its record only; the per-agent kernel - the home of fp64 - turns it into
        // an angle: EDGE_HEADING_REC)
        psi = __hiloint2double(__float_as_int(q.w), __float_as_int(q.z));
    }
    const unsigned at = atomicAdd(d.edge_n, 1u) % EDGE_CAP;
    EdgeRec *const o = d.edge + at;          // (member by member: a local EdgeRec went through scratch memory)
    const int32_t fl = (int32_t)((seen ? EDGE_SEEN : 0) | (side ? EDGE_SIDE : 0) | (d.state_current ? 0 : EDGE_HEADING_REC));
    const int32_t nx = atomicExch(&d.edge_head[a_recv], (int)at + 1);
    if (d.part4 != nullptr) {            // read by a per-agent wave of another XCD within this launch: write-through stores
#define CSF_ST_PUB(member, value) __hip_atomic_store(&o->member, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
        CSF_ST_PUB(xi, xi); CSF_ST_PUB(yi, yi); CSF_ST_PUB(psi, psi); CSF_ST_PUB(hfov, hfov);
        CSF_ST_PUB(fx, fx); CSF_ST_PUB(fy, fy); CSF_ST_PUB(fx2, fx2); CSF_ST_PUB(fy2, fy2);
        CSF_ST_PUB(recv, a_recv); CSF_ST_PUB(stamp, d.edge_stamp); CSF_ST_PUB(flags, fl); CSF_ST_PUB(next, nx);
#undef CSF_ST_PUB
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the entry before the bit that announces it
    } else {
        o->xi = xi;
        o->yi = yi;
        o->psi = psi;
        o->hfov = hfov;
        o->fx = fx;
        o->fy = fy;
        o->fx2 = fx2;
        o->fy2 = fy2;
        o->recv = a_recv;
        o->stamp = d.edge_stamp;
        o->flags = fl;
        o->next = nx;
    }
    atomicOr(&d.status[a_recv], CSF_ST_EDGE);
}

// vehicle.py:1560-1648: force of source (record q) on receiver r, returned as magnitude F and an
// unnormalised direction (gx, gy) with F already holding 1/|g|.  (dx, dy) = receiver - source.
// SGN0: np.sign(phi) with sign(0) = 0, as the reference has it; false: the sign bit of sin(phi), bit for bit what the packed
// field (field_twod_x2) computes - for the caller that takes a fast-path evaluation back.
// sgf: +1 / -1 puts that in place of np.sign(phi) (0: the pair's own) - for a pair whose sign is decided elsewhere.
template <bool SGN0 = true>
__device__ __forceinline__ void field_twod(const PairConsts &k, const Recv &r, const float4 q, float dx,
                                           float dy, float r2, float &F, float &gx, float &gy, float sgf = 0.0f) {
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
    if (sgf != 0.0f) sg = sgf;
    bool pos = cphi >= 0.0f;
    float bs = (SGN0 || sgf != 0.0f) ? big * sg : __builtin_copysignf(big, sphi), al = 0.5f * sphi * rs;
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
// NEARFLAG: near0 / near1 (wave masks) report the pairs closer than k.rnear (pair_cull_kernel corrects them from the precise
// records).
template <bool FULL, bool NEARFLAG = false>
__device__ __forceinline__ void field_twod_x2(const PairConsts &k, const Recv &r, const v2f qx, const v2f qy,
                                              const v2f qc, const v2f qs, bool valid0, bool valid1, float &ax,
                                              float &ay, unsigned long long *near0 = nullptr, unsigned long long *near1 = nullptr) {
    v2f dx = r.x - qx, dy = r.y - qy;                 // vehicle.py:1615-1616
    v2f r2 = dx * dx + dy * dy;
    if (NEARFLAG) {   // (every mask the result of ONE compare: csf_pair.hip ballot1)
        *near0 = __builtin_amdgcn_ballot_w64(r2.x < k.rnear2);
        *near1 = __builtin_amdgcn_ballot_w64(r2.y < k.rnear2);
        if (!FULL) *near0 &= __builtin_amdgcn_ballot_w64(valid0), *near1 &= __builtin_amdgcn_ballot_w64(valid1);
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
    // np.sign(phi) (:1625) as the sign BIT of sin(phi) (two v_bfi; sign(0) = 0 does not come out of it).  A pair within fp32
    // rounding of phi = 0 - |sin(phi)| below the band of the sine of a bearing at a metre (csf_engine.hip: set_fov_band), which
    // holds phi = 0 itself - is noted like a near pair (NEARFLAG) and decided where those are corrected (csf_pair.hip: near_drain)
    if (NEARFLAG) {
        const float Ts = k.fovT0 + k.fovT1;           // (|al| <= 0.71 |sin phi|: nothing inside the band is missed)
        unsigned long long s0 = __builtin_amdgcn_ballot_w64(__builtin_fabsf(al.x) <= Ts), s1 = __builtin_amdgcn_ballot_w64(__builtin_fabsf(al.y) <= Ts);   // (<=: phi = 0 itself with the band switched off)
        if (!FULL) s0 &= __builtin_amdgcn_ballot_w64(valid0), s1 &= __builtin_amdgcn_ballot_w64(valid1);
        *near0 |= s0;
        *near1 |= s1;
    }
    v2f bs{__builtin_copysignf(big.x, sphi.x), __builtin_copysignf(big.y, sphi.y)};    // sign(phi) big          :1625
    const bool p0 = cphi.x >= 0.0f, p1 = cphi.y >= 0.0f;
    v2f h1{p0 ? __builtin_fabsf(al.x) : big.x, p1 ? __builtin_fabsf(al.y) : big.y};   // |sin phi| hrs      :1624
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
                                        const v2f qs, bool &k0, bool &k1, bool &m0, bool &m1) {
    const v2f dx = r.x - qx, dy = r.y - qy;
    const v2f r2 = dx * dx + dy * dy;
    const v2f X = dx * qc + dy * qs;
    const v2f srel = qs * r.c - qc * r.s;
    const v2f s2 = srel * srel;
    const v2f e = k.e0 - k.e1 * s2, A = k.tA0 + k.tA1 * s2, B = k.tB0 + k.tB1 * s2;
    const v2f inv = rsq2(r2);
    const v2f c = X * inv;                             // r2 = 0 (the receiver itself): NaN
    const v2f S = A + B * c;
    const v2f eX = e * X;
    const v2f f = (r2 - eX * eX) - S * S;              // > 0: out of reach
    m0 = m1 = false;
    if (!FOV) {
        k0 = !(f.x > 0.0f);                            // (batches wholly inside the field of view never hold the receiver itself)
        k1 = !(f.y > 0.0f);
    } else {
        // Field of view in the cosine of the bearing, c = t / rho against cos(hfov / 2), with the band T of its fp32 rounding
        // (csf_engine.hip: set_fov_band): a source is kept when c - ch > -T - tracked, or within rounding of the edge: MARGINAL,
        // decided exactly where the pair is corrected - and marginal when c - ch < T as well.  The rsq is the reach test's.
        // Each result is ONE compare (a wave mask straight from the instruction; the `and` of two compares would go through a
        // vector register on its way to a ballot): reach and field of view are merged with a min, the receiver itself
        // (r2 = 0: NaN in both) fails the compare.  m0 / m1 are meaningful for kept sources only.
        const v2f t = -(dx * r.c + dy * r.s);
        const v2f T = k.fovT0 + k.fovT1 * inv;
        const v2f dc = t * inv - k.chk;
        const v2f z = dc + T;
        if (!P2R) {
            k0 = __builtin_fminf(-f.x, z.x) > 0.0f;        // in reach (f < 0; NaN: the other operand) and c - ch > -T
            k1 = __builtin_fminf(-f.y, z.y) > 0.0f;
        }
        m0 = dc.x < T.x;
        m1 = dc.y < T.y;
        if (P2R) {
            const v2f ss = (r.s * dx - r.c * dy) * inv;    // sin(bearing) > 0: the source is to the left; same band
            k0 = (!(f.x > 0.0f)) & (z.x > 0.0f) & !(ss.x > T.x);
            k1 = (!(f.y > 0.0f)) & (z.y > 0.0f) & !(ss.y > T.y);
            m0 = m0 | (ss.x > -T.x);
            m1 = m1 | (ss.y > -T.y);
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
