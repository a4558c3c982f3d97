// csf_tick.hip — the whole tick of a SMALL population in one launch, k ticks per launch (gfx950).
//
// SocialForceIntersection.step() (intersection.py:866-896) is, for the populations the reference's own users run
// (3 ... a few hundred cyclists), two kernel launches at their floor in the general path: pair kernel + per-agent
// kernel, 10-20 us per tick of which less than 2 us is arithmetic.  Here ONE persistent launch runs n_ticks ticks:
//
//   every wave owns R receivers (R = 1 up to 1024 road users); per tick the workgroup stages all source records in
//   LDS, every wave sums the field over the sources for its receivers (sources in the lanes, cull first: field-of-view
//   test -> LDS queue -> packed field on full batches, exactly the functions of csf_field.h), lane r of the wave then
//   runs the per-agent tick of receiver r (csf_agent_dev.h: destination force, clamp + combine, controller + kinematics)
//   on registers that stay loaded between ticks, and publishes the road user's next source record;
//   a device-scope barrier (one atomic counter; skipped for a single workgroup) separates the ticks.
//
// The exchanged source records are double-buffered by tick parity and carry the position as TWO floats (hi + lo: the
// fp64 position to ~1e-14 relative), so that receiver - source is formed as (hi - hi) + (lo - lo): exact to fp32 of the
// DISTANCE, as the reference's fp64 difference (vehicle.py:1615-1617) - no near-pair path is needed here.
// The canonical arrays of the engine (state, records, pointers ...) are written when the launch ends.
//
// Not here (the engine then takes the general path): several parameter sets, shards, the opt-in history ring, replay.
#include <type_traits>

#include "csf_agent_dev.h"
#include "csf_field.h"

namespace csf {

constexpr int TK_WAVES = 4, TK_BLOCK = TK_WAVES * WAVE;
constexpr int TK_QCAP = 256;

// one workgroup waits for all: arrival counter in device memory (monotone over the life of the engine: the host
// passes the count it has reached, target = that + workgroups x barriers passed).  One thread per workgroup does the
// device-scope part: the workgroup barrier in front of it orders the other threads' stores before its release.
// Bounded spin: a grid that cannot be resident as a whole (it always can: <= 256 workgroups of 256 threads) must not
// hang the device - after ~0.5 s every workgroup gives up and the host reports the error.
__device__ __forceinline__ bool grid_barrier(unsigned long long *ctr, unsigned long long target) {
    __shared__ int ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // (release: the workgroup's records first)
        int good = 1;
        long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__hip_atomic_load(ctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull || ++spins > (1L << 24)) {
                __hip_atomic_store(ctr + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // everybody leaves
                good = 0;
                break;
            }
        }
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

// FIELD 0: the TwoD field (vehicle.py:1560-1648), FIELD 1: the Bicycle field (vehicle.py:1054-1147).  NS: the LDS tile
// holds up to NS sources.
template <int MODEL, bool P2R, int NS>
__global__ __launch_bounds__(TK_BLOCK, 1) void tick_kernel(const Dev d, const TickArgs t) {
    constexpr bool BIKE = MODEL == CSF_BICYCLE;
    __shared__ float sx[NS], sy[NS], sc[NS], ss[NS], sxl[NS], syl[NS];
    __shared__ float2 sb[BIKE ? NS : 1];
    __shared__ unsigned short queue[TK_WAVES][TK_QCAP];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int R = t.recv_per_wave;
    const int64_t a0 = ((int64_t)blockIdx.x * TK_WAVES + wave) * R;     // first receiver (slot) of this wave
    const int64_t n = d.n, cap = d.cap;
    const int n64 = (int)((n + WAVE - 1) / WAVE * WAVE);

    // ---- the road user of this lane (lanes < R), loaded once: vehicle.s, queue pointer, navigation state ... ------
    const int64_t a = a0 + lane;
    const bool mine = lane < R && a < n && d.alive[a] != 0;
    const int64_t am = mine ? a : 0;
    Agent g;
    g.a = am;
    g.x = d.s[am]; g.y = d.s[cap + am]; g.psi = d.s[2 * cap + am]; g.v = d.s[3 * cap + am];
    g.delta = d.s[4 * cap + am]; g.theta = d.s[5 * cap + am];
    g.vdes = d.vdes[am];
    g.qb = d.qbeg[am]; g.K = d.qlen[am]; g.ptr = d.ptr[am];
    g.zn = d.znav[am] & 3;
    g.zv0 = d.znp[am]; g.zd0 = d.znp[cap + am]; g.zd1 = d.znp[2 * cap + am];
    g.ti = d.ti[am];
    g.st = d.status[am];
    g.cs_fresh = false;
    g.p = &d.p;
    g.pb = d.pb;
    double Fx = 0, Fy = 0, fdx = 0, fdy = 0, frx = 0, fry = 0;

    // the source record of a road user as it is exchanged between the waves
    auto publish = [&](int buf) {
        if (lane < R && a < n64) {
            float4 A = make_float4(1e15f, 1e15f, 1.0f, 0.0f);          // dead slot / padding: the sentinel
            float2 B = make_float2(0.f, 0.f), Cq = make_float2(0.f, 1.f);
            if (mine) {
                double sn, cs;
                if (g.cs_fresh) sn = g.spsi, cs = g.cpsi;
                else sincos(g.psi, &sn, &cs);
                const double X = g.x - d.ox, Y = g.y - d.oy;
                A = make_float4((float)X, (float)Y, (float)cs, (float)sn);
                B = make_float2((float)(X - (double)A.x), (float)(Y - (double)A.y));
                if (BIKE) {                                            // vehicle.py:1062-1064 (v <= 0: e := 0)
                    const double e = g.v > 0.0 ? fmin(pow(g.v / d.p.v_max_riding[1], 0.1), 0.7) : 0.0;
                    Cq = make_float2((float)e, (float)(1.0 / sqrt(1.0 - e * e)));
                }
            }
            t.xa[buf][a] = A;
            t.xb[buf][a] = B;
            if (BIKE) t.xc[buf][a] = Cq;
        }
    };
    // (the padding slots behind the last wave's receivers: written by whoever covers them - the grid covers [0, n64))
    publish(0);
    unsigned long long passed = 0;
    const bool one = gridDim.x == 1;
    if (one) __syncthreads();
    else if (!grid_barrier(t.barrier, t.barrier_base + gridDim.x * ++passed)) return;

    PairConsts k = d.pc;
    for (int64_t tick = 0; tick < t.n_ticks; tick++) {
        const int cur = (int)(tick & 1);
        // ---- 1. every source record of the population -> LDS -------------------------------------------------------
        for (int i = threadIdx.x; i < n64; i += TK_BLOCK) {
            const float4 A = t.xa[cur][i];
            const float2 B = t.xb[cur][i];
            sx[i] = A.x; sy[i] = A.y; sc[i] = A.z; ss[i] = A.w; sxl[i] = B.x; syl[i] = B.y;
            if (BIKE) sb[i] = t.xc[cur][i];
        }
        __syncthreads();
        // ---- 2. column sums of this wave's receivers (intersection.py:814-843) -------------------------------------
        float myrx = 0.f, myry = 0.f, myroadx = 0.f, myroady = 0.f;
        for (int r = 0; r < R; r++) {
            const int64_t ar = a0 + r;                                  // (uniform)
            if (ar >= n) break;
            const int ir = (int)ar;
            Recv ru{sx[ir], sy[ir], sc[ir], ss[ir]};
            const float rxl = sxl[ir], ryl = syl[ir];
            float ax = 0.f, ay = 0.f;
            if (d.n_live > 1) {
                if (BIKE) {      // ~30 instructions per pair: evaluated lane by lane, masked (no queue)
                    for (int i = lane; i < n64; i += WAVE) {
                        const float dx = (ru.x - sx[i]) + (rxl - sxl[i]), dy = (ru.y - sy[i]) + (ryl - syl[i]);
                        float r2 = dx * dx + dy * dy;
                        const bool in = tracked<P2R>(k.chs, ru, dx, dy, r2);
                        r2 = fmaxf(r2, 1e-30f);
                        float F, gx, gy;
                        field_bicycle(k, make_float4(0.f, 0.f, sc[i], ss[i]), sb[i], dx, dy, r2, F, gx, gy);
                        F = in ? F : 0.0f;
                        ax += F * gx;
                        ay += F * gy;
                    }
                } else {         // cull first: test -> queue -> packed field on full batches of 128
                    int qlen = 0, qhead = 0;
                    auto pop = [&](auto full) {
                        constexpr bool FULL = decltype(full)::value;
                        const int m = FULL ? 128 : qlen;
                        const bool v0 = lane < m, v1 = lane + WAVE < m;
                        int i0 = queue[wave][(qhead + lane) & (TK_QCAP - 1)], i1 = queue[wave][(qhead + WAVE + lane) & (TK_QCAP - 1)];
                        if (!FULL) i0 = v0 ? i0 : ir, i1 = v1 ? i1 : ir;
                        field_twod_x2<FULL, false, true>(k, ru, v2f{sx[i0], sx[i1]}, v2f{sy[i0], sy[i1]}, v2f{sc[i0], sc[i1]},
                                                         v2f{ss[i0], ss[i1]}, v0, v1, ax, ay, nullptr, nullptr,
                                                         v2f{rxl - sxl[i0], rxl - sxl[i1]}, v2f{ryl - syl[i0], ryl - syl[i1]});
                        qhead = (qhead + m) & (TK_QCAP - 1);
                        qlen -= m;
                    };
                    for (int i = lane; i < n64; i += WAVE) {
                        const float dx = (ru.x - sx[i]) + (rxl - sxl[i]), dy = (ru.y - sy[i]) + (ryl - syl[i]);
                        const bool in = tracked<P2R>(k.chs, ru, dx, dy, dx * dx + dy * dy);
                        const unsigned long long m = __ballot(in);
                        if (in) {
                            const int pre = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                            queue[wave][(qhead + qlen + pre) & (TK_QCAP - 1)] = (unsigned short)i;
                        }
                        qlen += __builtin_popcountll(m);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        if (qlen >= 128) pop(std::true_type{});
                    }
                    if (qlen > 0) pop(std::false_type{});
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    ax += __shfl_xor(ax, o, WAVE);
                    ay += __shfl_xor(ay, o, WAVE);
                }
            }
            float qx = 0.f, qy = 0.f;
            if (d.nv > 0) {      // RoadEdge.calcRepulsiveForce (intersection.py:226-242): vertices in the lanes
                for (int64_t iv = lane; iv < d.nv_pad; iv += WAVE) {
                    const float4 vtx = d.rv[iv];                        // (offset from the tile's origin, -F0, -(sigma + 1) / 2)
                    const float2 ot = d.rvo[iv >> 10];
                    const float ex = vtx.x - ((ru.x - ot.x) + rxl), ey = vtx.y - ((ru.y - ot.y) + ryl);
                    const float r2 = ex * ex + ey * ey;
                    const float lg = fminf(vtx.w * fast_log2(r2), 120.0f);   // r = 0: finite, times ex = ey = 0
                    const float m = fast_exp2(lg) * vtx.z;
                    qx += m * ex;
                    qy += m * ey;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    qx += __shfl_xor(qx, o, WAVE);
                    qy += __shfl_xor(qy, o, WAVE);
                }
            }
            if (lane == r) myrx = ax, myry = ay, myroadx = qx, myroady = qy;
        }
        // ---- 3. the per-agent tick of lane r's road user, on registers (csf_agent_dev.h) -------------------------
        if (mine) {
            dest_force<MODEL>(d, g, fdx, fdy);                          // vehicle.calcDestinationForce
            frx = 0, fry = 0;
            Fx = fdx, Fy = fdy;
            if (d.n_live > 1) {                                         // intersection.py:813, 825, 849-851
                frx = (double)myrx, fry = (double)myry;
                const double rin = sqrt(frx * frx + fry * fry), lim = sqrt(fdx * fdx + fdy * fdy);
                if (rin > lim) {                                        // utils.py:79-84
                    frx = frx * lim / rin;
                    fry = fry * lim / rin;
                }
                Fx = frx + fdx;                                         // :847-848
                Fy = fry + fdy;
            }
            if (d.nv > 0) {                                             // :854-857
                Fx += (double)myroadx;
                Fy += (double)myroady;
            }
            if (!(isfinite(Fx) && isfinite(Fy))) g.st |= CSF_ST_NAN;
            g.cs_fresh = false;
            integrate<MODEL>(d, g, Fx, Fy);                             // vehicle.step
        }
        publish(cur ^ 1);
        if (one) __syncthreads();
        else if (!grid_barrier(t.barrier, t.barrier_base + gridDim.x * ++passed)) return;
    }
    // ---- the engine's own arrays, as the general path leaves them after a tick ------------------------------------
    if (mine) {
        d.s[a] = g.x; d.s[cap + a] = g.y; d.s[2 * cap + a] = g.psi; d.s[3 * cap + a] = g.v;
        d.s[4 * cap + a] = g.delta; d.s[5 * cap + a] = g.theta;
        d.ptr[a] = g.ptr;
        d.znav[a] = (uint8_t)g.zn;
        d.znp[a] = g.zv0; d.znp[cap + a] = g.zd0; d.znp[2 * cap + a] = g.zd1;
        d.ti[a] = g.ti;
        d.status[a] = g.st;
        d.F[a] = Fx; d.F[cap + a] = Fy; d.F[2 * cap + a] = fdx; d.F[3 * cap + a] = fdy; d.F[4 * cap + a] = frx; d.F[5 * cap + a] = fry;
        write_record(d, d.p, a, d.rorg[a], g.x, g.y, g.psi, g.v, g.cs_fresh, g.cpsi, g.spsi);
    }
}

// receivers per wave and workgroups for a population of n slots: one receiver per wave up to 1024 slots
void tick_layout(int64_t n, int *recv_per_wave, int *blocks) {
    int R = (int)((n + 1023) / 1024);
    if (R < 1) R = 1;
    const int64_t n64 = (n + WAVE - 1) / WAVE * WAVE;
    const int64_t waves = (n64 + R - 1) / R;                            // (the padding up to n64 is published by the grid too)
    *recv_per_wave = R;
    *blocks = (int)((waves + TK_WAVES - 1) / TK_WAVES);
}

int tick_blocks(const Dev &d) {
    int R, blocks;
    tick_layout(d.n, &R, &blocks);
    return blocks;
}

bool tick_fits(const Dev &d) {
    int R, blocks;
    tick_layout(d.n, &R, &blocks);
    return d.n >= 1 && d.n <= TICK_MAX_AGENTS && blocks <= 256 && R <= WAVE && d.p.model != CSF_UNCONTROLLED;
}

template <int MODEL>
static void launch_tick_model(const Dev &d, const TickArgs &t, int blocks, hipStream_t st, hipEvent_t t0, hipEvent_t t1) {
    const dim3 g((unsigned)blocks), b(TK_BLOCK);
    if (d.n <= 1024) {
        if (d.pc.p2r) hipExtLaunchKernelGGL((tick_kernel<MODEL, true, 1024>), g, b, 0, st, t0, t1, 0, d, t);
        else hipExtLaunchKernelGGL((tick_kernel<MODEL, false, 1024>), g, b, 0, st, t0, t1, 0, d, t);
    } else {
        if (d.pc.p2r) hipExtLaunchKernelGGL((tick_kernel<MODEL, true, TICK_MAX_AGENTS>), g, b, 0, st, t0, t1, 0, d, t);
        else hipExtLaunchKernelGGL((tick_kernel<MODEL, false, TICK_MAX_AGENTS>), g, b, 0, st, t0, t1, 0, d, t);
    }
}

void launch_tick(const Dev &d, TickArgs t, hipStream_t st, hipEvent_t t0, hipEvent_t t1) {
    int blocks;
    tick_layout(d.n, &t.recv_per_wave, &blocks);
    switch (d.p.model) {
    case CSF_BICYCLE: launch_tick_model<CSF_BICYCLE>(d, t, blocks, st, t0, t1); break;
    case CSF_TWOD: launch_tick_model<CSF_TWOD>(d, t, blocks, st, t0, t1); break;
    case CSF_INVPEND: launch_tick_model<CSF_INVPEND>(d, t, blocks, st, t0, t1); break;
    case CSF_PLANARBIKE: launch_tick_model<CSF_PLANARBIKE>(d, t, blocks, st, t0, t1); break;
    default: launch_tick_model<CSF_PLANARPOINT>(d, t, blocks, st, t0, t1); break;
    }
}

}  // namespace csf
