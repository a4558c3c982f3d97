// csf_agent.hip — fused per-agent tick (fp64, one lane per agent, SoA state => coalesced).
//
// Replaces, per tick and per agent:
//   PH_DEST       vehicle.calcDestinationForce: destination queue (vehicle.py:545-594), navigation state
//                 machine (:354-457), straight-line force (:1150-1194, 2078-2108) or spline path planner
//                 (:1416-1558; scipy splprep/splev restated as a closed 4..6-point cubic B-spline);
//   PH_COMBINE    column-sum tail of calc_forces: limitMagnitude clamp to |F_dest|, + F_dest, + road
//                 edges, store vehicle.force (intersection.py:841-862; utils.py:56-86);
//   PH_INTEGRATE  vehicle.step: controller + kinematics (vehicle.py:1218-1289, 1386-1414, 1810-1950;
//                 dynamics.py:996-1079), ring-buffer bookkeeping, and the refreshed (x, y, psi) snapshot
//                 (intersection.py:660-677) written as the fp32 source record of the next tick.
// The O(N) work is done in fp64 so that the only fp32 rounding in a tick is the pair sum.
#include <algorithm>

#include "csf_agent_dev.h"
#include "csf_field.h"

namespace csf {

template <int MODEL, bool HET = false>
__global__ __launch_bounds__(256) void agent_kernel(const Dev d, const int phases) {
    const uint32_t ka_lines = kernarg_touch<(int)sizeof(Dev) + 4>();   // (csf_dev.h)
    const int64_t a = d.lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // CSF_TRACE_AGENT (measurement aid, tools/agent_timeline.py): where does a wave's time go?  Every stamp waits for what
    // was issued before it, so the traced kernel is a little slower than the product's.
    uint64_t *const tr = d.atrace ? d.atrace + 8 * ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) : nullptr;
    if (tr != nullptr && (threadIdx.x & 63) == 0) tr[0] = wall_clock64();
    agent_body<MODEL, HET, false>(d, phases, a, tr, ka_lines, 0.0, 0.0);
}

// The per-agent tick BESIDE the pair launch that feeds it (csf_dev.h: part4; engine/tick.inc: enqueue_chase_tick): the same
// code with the sums waited for between the destination-force phase and the rest (agent_body<.., MID = 3>).
template <int MODEL>
// (`phases` is an argument, always all three: with a constant the compiler contracts the fp64 chains of some rider classes differently, and the
// two paths must agree to the last bit)
__global__ __launch_bounds__(256) void agent_chase_kernel(const Dev d, const int phases) {
    const uint32_t ka_lines = kernarg_touch<(int)sizeof(Dev) + 4>();
    const int64_t a = d.lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t *const tr = d.atrace ? d.atrace + 8 * ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) : nullptr;
    if (tr != nullptr && (threadIdx.x & 63) == 0) tr[0] = wall_clock64();
    unsigned long long *const ck = d.chase_clock ? d.chase_clock + 8 * d.chase_slot : nullptr;
    if (ck != nullptr && (threadIdx.x & 63) == 0) atomicMin(ck + 4, (unsigned long long)wall_clock64());
    agent_body<MODEL, false, false, 3>(d, phases, a, tr, ka_lines, 0.0, 0.0);
    if (ck != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((threadIdx.x & 63) == 0) atomicMax(ck + 6, (unsigned long long)wall_clock64());
    }
}

// One wave that holds the per-agent launch behind it (same stream) back until `want` pair workgroups are through: launched at once the
// 256 per-agent waves would sit on their registers for the whole pair launch and cost it a workgroup per CU.
__global__ __launch_bounds__(64) void chase_gate_kernel(const unsigned *through, unsigned want, unsigned *gave_up, unsigned *gave_up_host, unsigned long long *ck) {
    if (threadIdx.x != 0) return;
    if (ck != nullptr) ck[2] = wall_clock64();
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(through, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
        if (++spins > CHASE_SPIN_LIMIT) {
            atomicAdd(gave_up, 1u);
            if (gave_up_host != nullptr) __hip_atomic_store(gave_up_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        __builtin_amdgcn_s_sleep(32);
    }
    if (ck != nullptr) ck[3] = wall_clock64();
}

__global__ void chase_sync_kernel(const Dev d, float4 *rec_o, float4 *recg_o, float4 *recs_o, float2 *rec2_o, double *cur, int64_t nrecg,
                                  unsigned *through);

// (a hardware queue gets its scratch memory when a kernel that needs some is first dispatched on it - the per-agent kernels spill ~200 bytes
// per lane: this one asks for as much, on the stream that has not run them yet)
__global__ __launch_bounds__(64) void chase_scratch_warm_kernel(unsigned *out, int k) {
    volatile unsigned buf[96];
    for (int i = 0; i < 96; i++) buf[i] = (unsigned)(i * k) + threadIdx.x;
    unsigned acc = 0;
    for (int i = 0; i < 96; i++) acc += buf[(i * 7 + k) % 96];
    if (acc == 0xFFFFFFFFu) out[63] = acc;     // (never, for the k it is called with: nothing is written)
}

void launch_chase_scratch_warm(unsigned *out, hipStream_t st) {
    hipLaunchKernelGGL(chase_scratch_warm_kernel, dim3(256), dim3(64), 0, st, out, 3);
}

// code objects are loaded when a kernel is first asked for: do that where the host waits anyway, not in the first side-by-side tick
void preload_chase_kernels() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, (const void *)chase_gate_kernel);
    (void)hipFuncGetAttributes(&a, (const void *)chase_sync_kernel);
    (void)hipFuncGetAttributes(&a, (const void *)agent_chase_kernel<CSF_TWOD>);
    (void)hipFuncGetAttributes(&a, (const void *)agent_chase_kernel<CSF_INVPEND>);
    (void)hipFuncGetAttributes(&a, (const void *)agent_chase_kernel<CSF_PLANARPOINT>);
    (void)hipGetLastError();
}

bool launch_agent_chase(const Dev &d, hipStream_t st, hipEvent_t t0, hipEvent_t t1) {
    if (d.hi <= d.lo || d.part4 == nullptr || d.n_classes > 1) return false;
    const dim3 g((unsigned)((d.hi - d.lo + 63) / 64)), b(64);
    switch (d.p.model) {
    case CSF_TWOD: break;
    case CSF_INVPEND: break;
    case CSF_PLANARPOINT: break;
    default: return false;
    }
    const int ph = PH_DEST | PH_COMBINE | PH_INTEGRATE;
    hipLaunchKernelGGL(chase_gate_kernel, dim3(1), dim3(64), 0, st, d.chase_misc, d.chase_gate, d.chase_misc + 1, d.chase_err,
                       d.chase_clock ? d.chase_clock + 8 * d.chase_slot : nullptr);
    switch (d.p.model) {
    case CSF_TWOD: hipExtLaunchKernelGGL((agent_chase_kernel<CSF_TWOD>), g, b, 0, st, t0, t1, 0, d, ph); break;
    case CSF_INVPEND: hipExtLaunchKernelGGL((agent_chase_kernel<CSF_INVPEND>), g, b, 0, st, t0, t1, 0, d, ph); break;
    default: hipExtLaunchKernelGGL((agent_chase_kernel<CSF_PLANARPOINT>), g, b, 0, st, t0, t1, 0, d, ph); break;
    }
    return true;
}

// ---- a handful of road users: the whole tick in one wave, any number of ticks in one launch ---------------------------------
// The reference's own scenarios (three cyclists at a crossing, scenarios/*.py; BASELINE config 1) are latency, not work: a pair
// launch and a per-agent launch of 5 - 6 us each, nearly all of it launch, teardown and first round trips.  Up to
// SMALL_MAX road users of one class (not the UncontrolledVehicle's) are ticked by ONE wave instead - lane = road user - with nothing between the
// phases but the wave's own program order, and csf_step(n) is one launch for all n ticks:
//   snapshot (x, y, psi) of every road user, staged in LDS                intersection.py:660-677
//   every source j of the lane's group in turn: receiver - source formed in fp64; the field of view and
//   np.sign(phi) decided in fp32 on that difference with the band of ITS rounding (csf_field.h: tracked_precise,
//   side_undecided - the predicates of the pair kernels' exact path) and, inside the band, as the reference decides them -
//   fp64 atan2 -> limitAngle -> angleDifference (csf_dev.h: untracked_exact_xy), acos -> limitAngle -> sign
//   (sign_phi_exact); the field in fp32, summed in fp64 (per group in source order, the groups pairwise)
//                                                                         intersection.py:690-745, 814-843; vehicle.py:1560-1648
//   the per-agent tick with that sum (agent_body<FUSED>)                  see the head of this file
// No records are binned, nothing is noted or handed over: every yes / no is settled on the spot.
template <int MODEL>
__global__ __launch_bounds__(64) void small_tick_kernel(const Dev d, const int n_ticks) {
    const uint32_t ka_lines = kernarg_touch<(int)sizeof(Dev) + 4>();
    // Lane = (receiver, source group): with P the power of two that holds the road users, lane % P is the receiver and lane / P
    // one of 64 / P groups that share the sources between them (source j belongs to group j % G) - all 64 lanes work on the
    // pair term whatever the population, and the groups' sums meet in lanes 0 .. n - 1, which then tick their road user.
    __shared__ double sx[SMALL_MAX], sy[SMALL_MAX], spsi[SMALL_MAX], scs[SMALL_MAX], ssn[SMALL_MAX];
    __shared__ float2 se[MODEL == CSF_BICYCLE ? SMALL_MAX : 1];   // Bicycle field: (e, 1 / sqrt(1 - e^2)) of every source (vehicle.py:1062-1064)
    const int lane = (int)threadIdx.x;
    const int n = (int)d.n;
    int P = 1;
    while (P < n) P <<= 1;
    const int G = WAVE / P, i = lane & (P - 1), grp = lane / P;
    const int64_t cap = d.cap;
    const bool live = i < n;
    const int64_t a = live ? i : 0;
    const PairConsts k = d.pc;
    const bool p2r = d.p.priority_rule == CSF_P2R;
    // road elements (intersection.py:226-242; the curve scenario's ~1 500 vertices): staged once per launch - they are static
    __shared__ float4 srv[SMALL_ROAD_MAX];
    const int nvp = (int)d.nv_pad;
    for (int v = lane; v < nvp; v += WAVE) srv[v] = d.rv[v];
    for (int t = 0; t < n_ticks; t++) {
        // (own stores of the previous tick: the lanes of the first group wrote them, in this wave: program order)
        const double x = d.s[a], y = d.s[cap + a], psi = d.s[2 * cap + a];
        double sp, cp;
        sincos(psi, &sp, &cp);
        if (lane < n) sx[lane] = x, sy[lane] = y, spsi[lane] = psi, scs[lane] = cp, ssn[lane] = sp;
        if (MODEL == CSF_BICYCLE && lane < n) {                    // (what write_record keeps in rec2 for the pair kernels)
            const double v = d.s[3 * cap + a];
            const double e = v > 0.0 ? fmin(pow(v / d.p.v_max_riding[1], 0.1), 0.7) : 0.0;
            se[lane] = make_float2((float)e, (float)(1.0 / sqrt(1.0 - e * e)));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const Recv r{0.f, 0.f, (float)cp, (float)sp};
        double rx = 0.0, ry = 0.0;
        for (int j = grp; j < n; j += G) {                         // (lanes of one group: the same j)
            const double xs = sx[j], ys = sy[j], ps = spsi[j];
            const double ex = x - xs, ey = y - ys;                 // vehicle.py:1615-1616
            // the receiver itself and a road user on the very same spot (D2) add nothing
            if (!live || j == i || (ex == 0.0 && ey == 0.0)) continue;
            const float dx = (float)ex, dy = (float)ey, r2 = fmaxf(dx * dx + dy * dy, 1e-30f);
            const float4 q = make_float4(0.f, 0.f, (float)scs[j], (float)ssn[j]);
            bool edge;
            bool seen = p2r ? tracked_precise<true>(k, k.chs, r, dx, dy, r2, edge) : tracked_precise<false>(k, k.chs, r, dx, dy, r2, edge);
            if (edge) seen = !untracked_exact_xy(xs, ys, x, y, psi, d.p.hfov, p2r);   // (one pair in a million)
            if (!seen) continue;
            int sg = 1;
            float F, gx, gy;
            if (MODEL == CSF_BICYCLE) {                             // vehicle.py:1054-1147: no jump at phi = 0
                field_bicycle(k, q, se[j], dx, dy, r2, F, gx, gy);
            } else {
                float sgf = 0.0f;                                   // 0: the sign of the fp32 sine
                if (side_undecided(k, q, dx, dy, r2)) {
                    sg = sign_phi_exact(xs, ys, ps, x, y);
                    sgf = sg < 0 ? -1.0f : 1.0f;
                }
                field_twod(k, r, q, dx, dy, r2, F, gx, gy, sgf);
            }
            double wx = (double)(F * gx), wy = (double)(F * gy);
            if (sg == 0) {                                          // phi = 0 exactly: no tangential part, |F| = P along the line
                const double Pm = sqrt(wx * wx + wy * wy), il = 1.0 / sqrt(ex * ex + ey * ey);
                wx = Pm * ex * il;
                wy = Pm * ey * il;
            }
            rx += wx;
            ry += wy;
        }
        // the groups' sums of a receiver, added pairwise in a fixed order: every lane of it ends with the total
        for (int o = P; o < WAVE; o <<= 1) {
            rx += __shfl_xor(rx, o, WAVE);
            ry += __shfl_xor(ry, o, WAVE);
        }
        if (nvp > 0) {
            // vertices as offsets from the origin of their tile of 1 024 (csf_dev.h: rv, rvo): the receiver's offset from it is formed
            // in fp64; the sum as road_kernel forms it (fp32, r^-(sigma+1) as a power of rsq(r^2) where every edge shares an integer sigma)
            float qx = 0.f, qy = 0.f;
            const double bx = x - d.ox, by = y - d.oy;
            for (int base = 0; base < nvp; base += 1024) {
                const float2 ot = d.rvo[base >> 10];
                const float rxo = (float)(bx - (double)ot.x), ryo = (float)(by - (double)ot.y);
                const int cnt = nvp - base < 1024 ? nvp - base : 1024;
                for (int u = grp; u < cnt; u += G) {
                    const float4 v = srv[base + u];                // (x, y, -F0, -(sigma + 1) / 2); padding has F0 = 0
                    const float ex = v.x - rxo, ey = v.y - ryo, r2 = ex * ex + ey * ey;
                    float m;
                    if (d.road_np) {
                        const float inv = fminf(fast_rsq(r2), 1e6f), i2 = inv * inv;   // r = 0: finite, times ex = ey = 0
                        m = d.road_np == 2 ? i2 : d.road_np == 3 ? i2 * inv : d.road_np == 4 ? i2 * i2 : d.road_np == 5 ? i2 * i2 * inv : i2 * i2 * i2;
                    } else {
                        m = fast_exp2(fminf(v.w * fast_log2(r2), 120.f));
                    }
                    m *= v.z;
                    qx = m * ex + qx;
                    qy = m * ey + qy;
                }
            }
            for (int o = P; o < WAVE; o <<= 1) {
                qx += __shfl_xor(qx, o, WAVE);
                qy += __shfl_xor(qy, o, WAVE);
            }
            if (lane < n) d.froad[lane] = make_float2(qx, qy);    // (agent_body reads it back: the same lane, program order)
        }
        __builtin_amdgcn_wave_barrier();                          // (the staged snapshot is read by every lane before it is renewed)
        if (lane < n) agent_body<MODEL, false, true>(d, PH_DEST | PH_COMBINE | PH_INTEGRATE, lane, nullptr, ka_lines, rx, ry);
    }
    // csf_step_get_tick: what snapshot_kernel would pack in a launch of its own (slots are the population order here)
    if (d.snap != nullptr && lane < n) {
        const int ns = d.ns;
        for (int c = 0; c < ns; c++) d.snap[(int64_t)lane * ns + c] = d.s[(int64_t)c * cap + lane];
        double *F = d.snap + (int64_t)n * ns;
        F[lane] = d.F[lane];
        F[n + lane] = d.F[cap + lane];
        int32_t *ptr = (int32_t *)(F + 2 * n);
        ptr[lane] = d.ptr[lane];
        uint8_t *zn = (uint8_t *)(ptr + n);
        const int z = d.znav[lane] & 3;
        zn[3 * lane + 0] = z == 0;
        zn[3 * lane + 1] = z == 1;
        zn[3 * lane + 2] = z == 2;
    }
}

void launch_small_tick(const Dev &d, int n_ticks, hipStream_t st, hipEvent_t t0, hipEvent_t t1) {
    if (n_ticks <= 0 || d.n <= 0) return;
#define CSF_SMALL(MODEL) hipExtLaunchKernelGGL((small_tick_kernel<MODEL>), dim3(1), dim3(64), 0, st, t0, t1, 0, d, n_ticks)
    switch (d.p.model) {
    case CSF_BICYCLE: CSF_SMALL(CSF_BICYCLE); break;
    case CSF_TWOD: CSF_SMALL(CSF_TWOD); break;
    case CSF_INVPEND: CSF_SMALL(CSF_INVPEND); break;
    case CSF_PLANARBIKE: CSF_SMALL(CSF_PLANARBIKE); break;
    case CSF_BALANCINGRIDER: CSF_SMALL(CSF_BALANCINGRIDER); break;
    default: CSF_SMALL(CSF_PLANARPOINT); break;
    }
#undef CSF_SMALL
}

// (re)build the fp32 records, the short position ring and the model side-state from the fp64 state:
// Vehicle.__init__ (vehicle.py:154-160, 1728-1736; dynamics.py:828) and update_road_user_positions (:660-677)
__global__ void records_kernel(const Dev d) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= d.n_pad) return;
    if (a >= d.n || !d.alive[a]) {  // sentinel: far away, contributes exactly 0 (exp2 underflow), never NaN
        d.rec[a] = make_float4(1e15f, 1e15f, 1.0f, 0.0f);
        if (a < d.cap) d.recg[a] = d.rec[a];
        if (d.has_bike) d.rec2[a] = make_float2(0.0f, 1.0f);
        if (d.xbuf != nullptr) {
            d.xbuf[2 * a] = d.rec[a];
            d.xbuf[2 * a + 1] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        }
        return;
    }
    write_record(d, d.ptab[d.cls[a]], a, d.rorg[a], d.s[a], d.s[d.cap + a], d.s[2 * d.cap + a], d.s[3 * d.cap + a]);
}

void launch_agent(const Dev &d, int phases, hipStream_t st, hipEvent_t t0, hipEvent_t t1) {
    if (d.hi <= d.lo) return;
    constexpr int bs = 64;   // one-wave workgroups: 256 waves on 256 CUs measured 0.6 us faster than 64 workgroups of 4
    dim3 g((unsigned)((d.hi - d.lo + bs - 1) / bs)), b(bs);
#define CSF_AGENT(MODEL)                                                                                             \
    if (d.n_classes > 1) hipExtLaunchKernelGGL((agent_kernel<MODEL, true>), g, b, 0, st, t0, t1, 0, d, phases);           \
    else hipExtLaunchKernelGGL((agent_kernel<MODEL, false>), g, b, 0, st, t0, t1, 0, d, phases)
    // every vehicle class of the population in turn (one, unless csf_set_param_classes installed sets of several classes;
    // the launches touch disjoint agents).  The time stamps bracket the first launch.
    for (int m = 0; m < 7; m++) {
        if (!(d.model_mask >> m & 1)) continue;
        switch (m) {
        case CSF_BALANCINGRIDER: CSF_AGENT(CSF_BALANCINGRIDER); break;
        case CSF_UNCONTROLLED: CSF_AGENT(CSF_UNCONTROLLED); break;
        case CSF_BICYCLE: CSF_AGENT(CSF_BICYCLE); break;
        case CSF_TWOD: CSF_AGENT(CSF_TWOD); break;
        case CSF_INVPEND: CSF_AGENT(CSF_INVPEND); break;
        case CSF_PLANARBIKE: CSF_AGENT(CSF_PLANARBIKE); break;
        default: CSF_AGENT(CSF_PLANARPOINT); break;
        }
        t0 = t1 = nullptr;
    }
#undef CSF_AGENT
}

// Vehicle.updateDestination (vehicle.py:545-594) and Vehicle.updateNavState(stop) (vehicle.py:354-457) on their own, for
// the listed agents: the host mirror's methods of the same name run them here (what & 1: queue pointer, what & 2: navigation
// state machine, desired speed and distance to vd_out / ddest_out).  stop[k] >= 0 overrides the stop flag of the current
// destination for this call, as the reference's explicit `stop` argument does.
__global__ void nav_kat_kernel(const Dev d, const int32_t *idx, int64_t m, int what, const int32_t *stop, double *vd_out,
                               double *ddest_out) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const int64_t a = idx[k], cap = d.cap;
    Agent g;
    g.a = a;
    g.x = d.s[a];
    g.y = d.s[cap + a];
    g.v = d.s[3 * cap + a];
    g.vdes = d.vdes[a];
    g.qb = d.qbeg[a];
    g.K = d.qlen[a];
    g.ptr = d.ptr[a];
    g.zn = d.znav[a] & 3;
    g.zv0 = d.znp[a];
    g.zd0 = d.znp[cap + a];
    g.zd1 = d.znp[2 * cap + a];
    g.st = d.status[a];
    agent_params<true>(d, a, g);                             // (the class table is there for one parameter set as well)
    load_rows(d, g);
    if (what & 1) {
        update_destination(d, g);
        d.ptr[a] = g.ptr;
    }
    if (what & 2) {
        const bool forced = stop != nullptr && stop[k] >= 0;
        if (forced) g.rs[0] = stop[k] ? 1.0 : 0.0;           // (the row in registers; the slab keeps its flag)
        double ddest;
        const double vd = update_nav(d, g, ddest);
        d.znav[a] = (uint8_t)g.zn;
        d.znp[a] = g.zv0;
        d.znp[cap + a] = g.zd0;
        d.znp[2 * cap + a] = g.zd1;
        d.status[a] = g.st;
        vd_out[k] = vd;
        ddest_out[k] = ddest;
    }
}

void launch_nav_kat(const Dev &d, const int32_t *idx, int64_t m, int what, const int32_t *stop, double *vd_out,
                    double *ddest_out, hipStream_t st) {
    if (m <= 0) return;
    hipLaunchKernelGGL(nav_kat_kernel, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, st, d, idx, m, what, stop, vd_out, ddest_out);
}

// csf_get_tick: everything the host mirror refreshes after a tick, packed for one transfer
__global__ void snapshot_kernel(const Dev d, double *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // road user i of the population order
    const int64_t n = d.n_live;
    if (i >= n) return;
    const int64_t a = d.order ? d.order[i] : i;
    const int ns = d.ns;
    for (int c = 0; c < ns; c++) out[i * ns + c] = d.s[(int64_t)c * d.cap + a];
    double *F = out + n * ns;
    F[i] = d.F[a];
    F[n + i] = d.F[d.cap + a];
    int32_t *ptr = (int32_t *)(F + 2 * n);
    ptr[i] = d.ptr[a];
    uint8_t *zn = (uint8_t *)(ptr + n);
    const int z = d.znav[a] & 3;
    zn[3 * i + 0] = z == 0;
    zn[3 * i + 1] = z == 1;
    zn[3 * i + 2] = z == 2;
}

void launch_snapshot(const Dev &d, double *out, hipStream_t st) {
    if (d.n_live <= 0) return;
    hipLaunchKernelGGL(snapshot_kernel, dim3((unsigned)((d.n_live + 255) / 256)), dim3(256), 0, st, d, out);
}

void launch_records(const Dev &d, hipStream_t st) {
    if (d.n_pad <= 0) return;
    hipLaunchKernelGGL(records_kernel, dim3((unsigned)((d.n_pad + 255) / 256)), dim3(256), 0, st, d);
}

// The other halves of the double buffers <- this tick's records and fp64 positions, and the gate's counter cleared: what five
// copy / fill calls did when the side-by-side tick was entered (once per re-binning: a launch and its gap each), in one launch.
__global__ void chase_sync_kernel(const Dev d, float4 *rec_o, float4 *recg_o, float4 *recs_o, float2 *rec2_o, double *cur, int64_t nrecg,
                                  unsigned *through) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.n_pad) {
        rec_o[i] = d.rec[i];
        if (i < nrecg) recg_o[i] = d.recg[i];
        if (d.recs_valid) recs_o[i] = d.recs[i];
        if (d.has_bike) rec2_o[i] = d.rec2[i];
    }
    if (i < 3 * d.cap) cur[i] = d.s[i];
    if (through != nullptr && i == 0) *through = 0u;
}

void launch_chase_sync(const Dev &d, float4 *rec_o, float4 *recg_o, float4 *recs_o, float2 *rec2_o, double *cur, int64_t nrecg, unsigned *through,
                       hipStream_t st) {
    const int64_t n = std::max<int64_t>(d.n_pad, 3 * d.cap);
    if (n <= 0) return;
    hipLaunchKernelGGL(chase_sync_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, rec_o, recg_o, recs_o, rec2_o, cur, nrecg, through);
}

// ---- population changes on the device ---------------------------------------------------------------------------------
// One launch applies a batch of changes that the engine collected since the last device call (csf_engine.hip:
// flush_pending).  The items are independent of each other - the host has already dropped a retirement whose slot is
// spawned into again, folded a new queue of a new road user into its spawn record, and kept only the last queue of a
// slot - so they run side by side: thread k takes one retirement, one spawn, one queue, or one queue row.  The records
// are read straight from pinned host memory.
//
// spawn: Vehicle.__init__ (vehicle.py:64-204, 1728-1736; dynamics.py:828) for a new road user placed into a free slot:
//   what csf_add_agents writes into the host mirror on the upload path, plus the fp32 record.  The slot keeps the
//   position of the binned order it had (pos[]) - the host picks slots whose place is in the tail of sentinels behind
//   the sorted batches (csf_engine.hip: rebin), so until the next re-binning the arrivals share a few tail batches.
// retire: remove_road_user (intersection.py:576-634): the slot is no longer integrated and its record becomes the
//   sentinel that contributes exactly nothing as a source.
// requeue: Vehicle.setDestinations (vehicle.py:606-647): the queue of a slot now lives at another place of the slab.
// The batch header travels as a kernel argument (one round trip over PCIe less than reading it from the pinned buffer).
// [b0, b1): the batches of the binned order that receive arrivals in this launch (the host knows the places of the
// sentinel tail it hands out, csf_engine.hip: rebin): the workgroup that finishes last recomputes their bounding circles,
// which saves the bounds_kernel launch in front of the next pair launch.
__global__ __launch_bounds__(256) void patch_kernel(const Dev d, const PatchHeader h, const char *base, unsigned *ticket,
                                                    int b0, int b1) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t cap = d.cap;
    if (k < h.n_retire) {
        const int64_t a = ((const int32_t *)(base + h.off_retire))[k];
        d.alive[a] = 0;
        d.cls[a] = 0;                                              // (its sentinel record is looked up in set 0 from now on)
        const float4 q = make_float4(1e15f, 1e15f, 1.0f, 0.0f);
        d.rec[a] = q;
        d.recg[a] = q;
        if (d.recs_valid) {
            d.recs[d.pos[a]] = q;
            d.recb[d.pos[a]] = q;
        }
        if (d.has_bike) {
            d.rec2[a] = make_float2(0.0f, 1.0f);
            if (d.recs_valid) d.recs2[d.pos[a]] = make_float2(0.0f, 1.0f);
        }
    } else if ((k -= h.n_retire) < h.n_spawn) {
        const SpawnRec r = ((const SpawnRec *)(base + h.off_spawn))[k];
        const int64_t a = r.slot;
        const csf_params &p = d.ptab[r.cls];
        d.cls[a] = (uint8_t)r.cls;
        double s[STATE_ROWS];
        for (int c = 0; c < STATE_ROWS; c++) s[c] = c < d.ns ? r.s[c] : 0.0;
        s[2] = limit_angle(s[2]);                                  // vehicle.py:154-155
        for (int c = 0; c < STATE_ROWS; c++) d.s[c * cap + a] = s[c];
        d.vdes[a] = r.vdes;
        d.qbeg[a] = r.qbeg;
        d.qlen[a] = r.qlen;
        d.ptr[a] = 0;
        d.znav[a] = 0;                                             // vehicle.py:188
        for (int c = 0; c < 3; c++) d.znp[c * cap + a] = 0.0;
        d.ti[a] = 0;                                               // vehicle.py:146
        d.hx[a] = s[0];                                            // traj[:, 0] = s (vehicle.py:159-160)
        d.hy[a] = s[1];
        const double delta = d.ns > 4 ? s[4] : 0.0, theta = d.ns > 5 ? s[5] : 0.0;
        d.lti[0 * cap + a] = delta;                                // vehicle.py:1728
        d.lti[1 * cap + a] = 0.0;
        d.lti[2 * cap + a] = theta;
        d.lti[3 * cap + a] = 0.0;
        d.lti[4 * cap + a] = s[2];
        d.zrid[a] = s[3] < p.v_max_walk ? 0 : 1;                   // vehicle.py:1732-1736
        d.dgood[a] = (-p.delta_max_walk < delta && p.delta_max_walk > delta) ? 1 : 0;
        d.ppsi[a] = s[2];                                          // dynamics.py:828, 987-993
        if (p.model == CSF_BALANCINGRIDER) {                       // dynamics.py:306-307, 350-371: (roll, steer, rates, yaw) mirrored
            d.lti[0 * cap + a] = s[5];
            d.lti[1 * cap + a] = -s[4];
            d.lti[2 * cap + a] = s[7];
            d.lti[3 * cap + a] = -s[6];
            d.lti[4 * cap + a] = -s[2];
            d.ppsi[a] = s[3];                                      // the speed its first gains belong to
        }
        d.slen[a] = 0;                                             // (no prescribed trajectory until csf_set_script gives one)
        for (int c = 0; c < 6; c++) d.F[c * cap + a] = 0.0;
        d.status[a] = 0;
        d.alive[a] = 1;
        // its own origin: where it starts, rounded to 1/4 m; the slot keeps its place of the binned order
        const float2 o = make_float2((float)(0.25 * rint(4.0 * (s[0] - d.ox))), (float)(0.25 * rint(4.0 * (s[1] - d.oy))));
        d.rorg[a] = o;
        write_record(d, p, a, o, s[0], s[1], s[2], s[3]);
    } else if ((k -= h.n_spawn) < h.n_requeue) {
        const QueueRec r = ((const QueueRec *)(base + h.off_requeue))[k];
        d.qbeg[r.slot] = r.qbeg;
        d.qlen[r.slot] = r.qlen;
        const int32_t ptr = r.mode == 1 ? 0 : d.ptr[r.slot];       // vehicle.py:642-645: reset rewinds the pointer
        d.ptr[r.slot] = ptr < r.qlen ? ptr : r.qlen - 1;
    } else if ((k -= h.n_requeue) < 3 * h.n_rows) {
        d.q[3 * h.q_top + k] = ((const double *)(base + h.off_rows))[k];
    }
    if (b1 <= b0) return;                                          // (uniform) no circles to renew
    __shared__ bool last;
    __threadfence();                                               // this thread's records, before the ticket
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    for (int bb = b0 + (int)(threadIdx.x >> 6); bb < b1; bb += (int)(blockDim.x >> 6)) batch_circle(d, bb, threadIdx.x & 63, 0.0f, d.bnd);
    if (threadIdx.x == 0) *ticket = 0;                             // for the next launch
}

void launch_patch(const Dev &d, const PatchHeader &h, const void *records, unsigned *ticket, int b0, int b1, int64_t items,
                  hipStream_t st) {
    if (items > 0)
        hipLaunchKernelGGL(patch_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, d, h, (const char *)records, ticket, b0, b1);
}

}  // namespace csf
