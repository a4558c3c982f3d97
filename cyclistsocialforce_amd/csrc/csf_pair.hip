// csf_pair.hip — all-pairs repulsive force field + field-of-view mask + column sum (gfx950).
//
// Replaces, per tick:  SocialForceIntersection.get_untracked_foes (intersection.py:690-745),
// the N calls of vehicle.calcRepulsiveForce (intersection.py:814-823; vehicle.py:1560-1648 or
// 1054-1147) and the column sum of intersection.py:841-843;  RoadEdge.calcRepulsiveForce
// (intersection.py:226-242) for the static-obstacle term.
//
// Mapping (CDNA4, 64-wide waves): SOURCES sit in the lanes, RECEIVERS are wave-uniform.  One wave owns
// RPW receivers whose record and every receiver-only term live in scalar registers; the workgroup streams
// the fp32 source records (x, y, cos psi, sin psi) through an LDS tile that all of its waves share; each
// lane accumulates its sources' contribution and one butterfly of wavefront shuffles per receiver forms
// the column sum.  Nothing is gathered, no atomics: the result is bit-reproducible.
// The math is trig-free: every angle of the reference enters only through sin/cos, which are dot and
// cross products of unit vectors here.
#include "csf_dev.h"

namespace csf {

constexpr int RPW = 4;               // receivers per wave
constexpr int WPB = 4;               // waves per workgroup
constexpr int TILE = 1024;           // source records per LDS tile (16 KiB)
constexpr int BLOCK = WPB * WAVE;

struct Recv {
    float x, y, c, s;
};

__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

// intersection.py:690-745 for receiver r and source (dx, dy) = receiver - source.
// The receiver ignores the source when the bearing of the source, relative to the receiver's heading,
// is outside +-hfov/2 (hfov of the SOURCE's class, :733-735; one class per engine), when it is to the
// left under priority-to-the-right, or when it is the receiver itself / coincident (rho = 0).
// WIDE: hfov/2 > pi/2 (cos < 0); P2R: priority to the right.  Both are launch-time template flags so that
// the inner loop carries no uniform branches.
template <bool WIDE, bool P2R>
__device__ __forceinline__ bool tracked(const PairConsts &k, const Recv &r, float dx, float dy, float r2) {
    float t = -(dx * r.c + dy * r.s);  // rho * cos(relative bearing)
    float t2 = t * t, lim = k.ch2 * r2;
    bool in = WIDE ? ((t >= 0.0f) | (t2 <= lim)) : ((t >= 0.0f) & (t2 >= lim));
    if (P2R) in = in & !((r.s * dx - r.c * dy) > 0.0f);  // rho * sin(relative bearing) > 0
    return in & (r2 > 0.0f);
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
    float h1 = (pos ? small : big);  // sqrt((1 - cos phi)/2)  :1624
    float h2s = pos ? bs : al;                        // sqrt((1 + cos phi)/2) * sign(phi)
    float sigma = sga - sgb * h1;                     // :1624
    float dsig = -0.5f * sgb * h2s;                   // :1625
    float ec = e * cphi;
    float q2 = 1.0f - ec * ec;
    float qq = fast_sqrt(q2);
    float isg = fast_rcp(sigma);
    // :1631-1642 with the positive factor P / (sigma^2 q) taken out of both polar components and the
    // rotation by phi1 written with rho*cos(phi1) = dx, rho*sin(phi1) = dy
    float grho = q2 * sigma;
    float gphi = e * ec * sphi * sigma - q2 * dsig;
    gx = grho * dx - gphi * dy;
    gy = grho * dy + gphi * dx;
    float ig = fast_rsq(gx * gx + gy * gy);
    float P = fast_exp2(k.lf0 - k.kexp * (rho * qq * isg));  // f_0 exp(-rho q / sigma)      :1628
    F = P * ig;                                       // :1644-1646: |F| = P
}

// vehicle.py:1054-1147: older elliptic field of base Bicycle; q2 = (e, 1/sqrt(1-e^2)) of the source.
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

template <bool BICYCLE, bool WIDE, bool P2R>
__global__ __launch_bounds__(BLOCK) void pair_kernel(const Dev d) {
    __shared__ float4 tile[TILE];
    __shared__ float2 tile2[BICYCLE ? TILE : 1];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t j0 = d.lo + ((int64_t)blockIdx.x * WPB + wave) * RPW;

    // source chunk of this workgroup (blockIdx.y), in units of 64 records
    const int64_t units = d.n_pad / WAVE;
    const int64_t per = (units + d.n_split - 1) / d.n_split;
    const int64_t ibeg = (int64_t)blockIdx.y * per * WAVE;
    int64_t iend = ibeg + per * WAVE;
    if (iend > d.n_pad) iend = d.n_pad;

    Recv r[RPW];
    float ax[RPW], ay[RPW];
#pragma unroll
    for (int u = 0; u < RPW; u++) {
        int64_t j = j0 + u < d.hi ? j0 + u : d.hi - 1;  // clamp: duplicates are not stored
        float4 q = d.rec[j];
        r[u].x = q.x;
        r[u].y = q.y;
        r[u].c = q.z;
        r[u].s = q.w;
        asm volatile("" : "+v"(r[u].x), "+v"(r[u].y), "+v"(r[u].c), "+v"(r[u].s));
        ax[u] = 0.0f;
        ay[u] = 0.0f;
    }
    PairConsts k = d.pc;
    asm volatile("" : "+v"(k.sg0), "+v"(k.sg1), "+v"(k.sg2), "+v"(k.sg3), "+v"(k.e0), "+v"(k.e1), "+v"(k.lf0),
                 "+v"(k.kexp), "+v"(k.ch2));

    for (int64_t base = ibeg; base < iend; base += TILE) {
        int cnt = (int)((iend - base) < TILE ? (iend - base) : TILE);  // multiple of 64
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += BLOCK) {
            tile[t] = d.rec[base + t];
            if (BICYCLE) tile2[t] = d.rec2[base + t];
        }
        __syncthreads();
        for (int t = lane; t < cnt; t += WAVE) {
            float4 q = tile[t];
            float2 qb = BICYCLE ? tile2[t] : make_float2(0.f, 0.f);
#pragma unroll
            for (int u = 0; u < RPW; u++) {
                float dx = r[u].x - q.x, dy = r[u].y - q.y;  // vehicle.py:1615-1616
                float r2 = dx * dx + dy * dy;
                bool in = tracked<WIDE, P2R>(k, r[u], dx, dy, r2);
                r2 = fmaxf(r2, 1e-30f);  // self / coincident pair: keep every intermediate finite (F is masked)
                float F, gx, gy;
                if (BICYCLE) field_bicycle(k, q, qb, dx, dy, r2, F, gx, gy);
                else field_twod(k, r[u], q, dx, dy, r2, F, gx, gy);
                F = in ? F : 0.0f;
                ax[u] += F * gx;
                ay[u] += F * gy;
            }
        }
    }

    // column sum (intersection.py:841-843): butterfly over the 64 lanes of the wave
#pragma unroll
    for (int u = 0; u < RPW; u++) {
        float sx = ax[u], sy = ay[u];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sx += __shfl_xor(sx, o, WAVE);
            sy += __shfl_xor(sy, o, WAVE);
        }
        if (lane == 0 && j0 + u < d.hi) d.part[(int64_t)blockIdx.y * d.cap + j0 + u] = make_float2(sx, sy);
    }
}

// intersection.py:226-242: F = sum_k -F0 r_k^-sigma (v_k - p)/r_k over the polyline vertices.
// Same mapping: vertices in the lanes, receivers wave-uniform.
__global__ __launch_bounds__(BLOCK) void road_kernel(const Dev d) {
    __shared__ float4 tile[TILE];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t j0 = d.lo + ((int64_t)blockIdx.x * WPB + wave) * RPW;
    float rx[RPW], ry[RPW], ax[RPW], ay[RPW];
#pragma unroll
    for (int u = 0; u < RPW; u++) {
        int64_t j = j0 + u < d.hi ? j0 + u : d.hi - 1;
        float4 q = d.rec[j];
        rx[u] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, q.x)));
        ry[u] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, q.y)));
        ax[u] = 0.f;
        ay[u] = 0.f;
    }
    for (int64_t base = 0; base < d.nv_pad; base += TILE) {
        int cnt = (int)((d.nv_pad - base) < TILE ? (d.nv_pad - base) : TILE);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += BLOCK) tile[t] = d.rv[base + t];
        __syncthreads();
        for (int t = lane; t < cnt; t += WAVE) {
            float4 v = tile[t];  // (x, y, -F0, -(sigma+1)/2); padding has F0 = 0
#pragma unroll
            for (int u = 0; u < RPW; u++) {
                float ex = v.x - rx[u], ey = v.y - ry[u];        // :235-236 (numerators)
                float r2 = ex * ex + ey * ey;                    // :231-234
                float m = v.z * fast_exp2(v.w * fast_log2(r2));  // -F0 r^-(sigma+1)    :238
                m = r2 > 0.f ? m : 0.f;
                ax[u] += m * ex;                                 // :239-240
                ay[u] += m * ey;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < RPW; u++) {
        float sx = ax[u], sy = ay[u];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sx += __shfl_xor(sx, o, WAVE);
            sy += __shfl_xor(sy, o, WAVE);
        }
        if (lane == 0 && j0 + u < d.hi) d.froad[j0 + u] = make_float2(sx, sy);
    }
}

// Known-answer entry: m independent (source, receiver) pairs through the same device functions.
__global__ void pair_kat_kernel(const Dev d, const float4 *src, const float2 *src2, const float4 *recv,
                                int64_t m, int apply_fov, float2 *out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const PairConsts k = d.pc;
    float4 q = src[t], rr = recv[t];
    Recv r{rr.x, rr.y, rr.z, rr.w};
    float dx = r.x - q.x, dy = r.y - q.y, r2 = dx * dx + dy * dy;
    bool in = r2 > 0.f;
    if (apply_fov) {
        if (k.ch >= 0.f) in = k.p2r ? tracked<false, true>(k, r, dx, dy, r2) : tracked<false, false>(k, r, dx, dy, r2);
        else in = k.p2r ? tracked<true, true>(k, r, dx, dy, r2) : tracked<true, false>(k, r, dx, dy, r2);
    }
    r2 = fmaxf(r2, 1e-30f);
    float F, gx, gy;
    if (d.p.model == CSF_BICYCLE) field_bicycle(k, q, src2[t], dx, dy, r2, F, gx, gy);
    else field_twod(k, r, q, dx, dy, r2, F, gx, gy);
    if (d.p.model != CSF_BICYCLE && k.f0_zero) F = 0.f;
    F = in ? F : 0.f;
    out[t] = make_float2(F * gx, F * gy);
}

static dim3 recv_grid(const Dev &d, int split) {
    int64_t nloc = d.hi - d.lo;
    int64_t per_block = (int64_t)WPB * RPW;
    return dim3((unsigned)((nloc + per_block - 1) / per_block), (unsigned)split, 1);
}

template <bool BICYCLE>
static void launch_pair_t(const Dev &d, dim3 g, hipStream_t st) {
    const bool wide = d.pc.ch < 0.f, p2r = d.pc.p2r != 0;
    if (wide) {
        if (p2r) hipLaunchKernelGGL((pair_kernel<BICYCLE, true, true>), g, dim3(BLOCK), 0, st, d);
        else hipLaunchKernelGGL((pair_kernel<BICYCLE, true, false>), g, dim3(BLOCK), 0, st, d);
    } else {
        if (p2r) hipLaunchKernelGGL((pair_kernel<BICYCLE, false, true>), g, dim3(BLOCK), 0, st, d);
        else hipLaunchKernelGGL((pair_kernel<BICYCLE, false, false>), g, dim3(BLOCK), 0, st, d);
    }
}

void launch_pair(const Dev &d, hipStream_t st) {
    if (d.hi <= d.lo) return;
    dim3 g = recv_grid(d, d.n_split);
    if (d.p.model == CSF_BICYCLE) launch_pair_t<true>(d, g, st);
    else launch_pair_t<false>(d, g, st);
}

void launch_road(const Dev &d, hipStream_t st) {
    if (d.hi <= d.lo || d.nv == 0) return;
    hipLaunchKernelGGL(road_kernel, recv_grid(d, 1), dim3(BLOCK), 0, st, d);
}

void launch_pair_kat(const Dev &d, const float4 *src, const float2 *src2, const float4 *recv, int64_t m,
                     int apply_fov, float2 *out, hipStream_t st) {
    if (m <= 0) return;
    hipLaunchKernelGGL(pair_kat_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, d, src, src2, recv,
                       m, apply_fov, out);
}

}  // namespace csf
