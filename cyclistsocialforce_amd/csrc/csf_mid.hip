// csf_mid.hip — the whole tick of a mid-size population (a few dozen to ~1 300 road users) in ONE launch.
//
// Replaces, per tick, the same reference code as csf_pair.hip + csf_agent.hip together: get_untracked_foes
// (intersection.py:690-745), the N calls of calcRepulsiveForce and the column sum (:814-843), the clamp and the road term
// (:841-857), and vehicle.step for every road user (:891-892) - SocialForceIntersection.step (:866-896).
//
// Between 33 and ~1 300 road users - BASELINE config 2 (1 024 TwoDBicycle), and what the reference itself runs under SUMO -
// the tick is latency, not work: the plain pair launch and the per-agent launch take 8 + 7 us at N = 1 024, of which ~9 us are
// the two launches' fixed cost (dispatch, first round trips, teardown).  Here ONE WORKGROUP of twelve waves (eight for the
// InvPendulum's per-agent code) owns a group of G road users (4 ... 32 slots, so that the grid is about one workgroup per CU)
// for the whole tick:
//
//   wave 0        the destination-force phase of the group's road users - queue, navigation state, planner: it needs no sums -
//                 (csf_agent_dev.h: agent_body, lane = road user) ...
//   the others    ... while they form the group's repulsive sums, item by item from an LDS counter: an item is one batch of 64 sources - sources in the lanes, straight from the records in memory, asked
//                 for one item ahead; no wave shares a source with another, so there is no tile to stage - against four
//                 receivers (csf_pair_dev.h: plain_pair_eval - the very code of pair_kernel: mask, field, precise records for
//                 near and marginal pairs, hand-overs of the undecidable ones), column sum by lane exchange, one partial per
//                 item and receiver to LDS;
//   barrier
//   wave 0        adds a receiver's partials in item order (fp64; whoever took which item: bit-reproducible) and runs the rest
//                 of the per-agent tick: hand-overs, clamp, road term, controller + kinematics, ring, next tick's records.
//
// Nothing leaves the workgroup between the phases: no partial sums in memory, no counters, no fences, no workgroup ever looks at
// another's progress (a first version split a group's sums over several workgroups and let the last one to arrive carry on:
// the arrival protocol - write-through stores, an agent-scope atomic, agent-scope loads - put 3 us of memory round trips on
// every group's critical path, and the per-agent code's 255 registers halved the residency of the pair workgroups: 13 us per
// tick at N = 64 ... 512 against 12.5 with two launches, 22 against 15 at N = 1 024; with __threadfence() 80).  What one
// launch per tick does cost is a DOUBLE BUFFER: a group writes its road users' next records while other groups still read
// this tick's, so the records (rec, recg, rec2) exist twice and the launch writes the half it does not read (Dev::rec_w ...);
// the fp64 positions that undecidable pairs are handed over with come from a snapshot of the tick's start (Dev::src64), which
// the per-agent phase renews for the next tick in the other half as well.
#include "csf_agent_dev.h"
#include "csf_pair_dev.h"

namespace csf {

// waves of a workgroup: twelve where the per-agent code of the vehicle class leaves room for three waves per SIMD (168 registers),
// else eight - one workgroup per CU either way, and one more wave per SIMD on the pair sums
__host__ __device__ constexpr int mid_waves(int model) { return model == CSF_INVPEND ? 8 : 12; }
constexpr int MID_GROUP_MAX = 32;             // road users (slots) of a group at most
constexpr int MID_ITEMS_MAX = 384;            // (receiver set, source batch) items of a group at most: 12 KB of partial sums

template <int MODEL, bool P2R>
__global__ __launch_bounds__(mid_waves(MODEL) * WAVE) void mid_tick_kernel(const Dev d) {
    constexpr int MID_WAVES = mid_waves(MODEL);
    constexpr int FIELD = MODEL == CSF_BICYCLE ? 1 : 0;
    constexpr bool BICYCLE = FIELD != 0;
    // one partial sum per ITEM - four receivers (a receiver set) against one batch of 64 sources -, [2 u + component]: the waves
    // take the items as they come (an LDS counter), and the sum over a receiver's items in item order does not depend on who
    // took which
    __shared__ float psum[MID_ITEMS_MAX][2 * RPW];
    __shared__ int next_item;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = d.mid_group, sets = G / RPW;
    const int nb = (int)((d.n_src - d.src_beg) >> 6), items = sets * nb;
    const int64_t g0 = d.lo + (int64_t)blockIdx.x * G;        // first slot of the group
    const int64_t a = g0 + lane;                              // (wave 0: lane = road user)
    uint32_t ka_lines = kernarg_touch<(int)sizeof(Dev) + 4>();   // (every wave: one workgroup per CU, and the pair waves' first
    kernarg_touched(ka_lines);                                   //  scalar loads sit in front of their first sources)
    // CSF_TRACE_AGENT (tools/mid_timeline.py): 16 stamps per workgroup - wave 0: entry, destination force done, its share of the
    // sums done, behind the barrier, end; wave 1: entry, first sources loaded, sums done; the last wave: sums done
    uint64_t *const tr = d.atrace ? d.atrace + 16 * (int64_t)blockIdx.x : nullptr;
    auto stamp = [&](int k) {
        if (tr != nullptr) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (lane == 0) tr[k] = wall_clock64();
        }
    };
    if (threadIdx.x == 0) next_item = 0;
    __syncthreads();
    if (wave == 0) {
        stamp(0);
        if (lane < G) agent_body<MODEL, false, false, 1>(d, PH_DEST, a, nullptr, ka_lines, 0.0, 0.0);
        stamp(1);
    } else {   // the pair sums: the other waves take items until there are none left
        if (wave == 1) stamp(4);
        PairConsts k = d.pc;
        asm volatile("" : "+v"(k.sg0), "+v"(k.sg1), "+v"(k.sg2), "+v"(k.sg3), "+v"(k.e0), "+v"(k.e1), "+v"(k.lf0), "+v"(k.kexp), "+v"(k.chs));
        auto claim = [&]() {
            int got = 0;
            if (lane == 0) got = atomicAdd(&next_item, 1);
            return __builtin_amdgcn_readfirstlane(got);
        };
        // the sources of an item: one record per lane, straight from memory (asked for one item ahead of the arithmetic)
        auto fetch = [&](int item, float4 &q, float2 &o, float2 &qb) {
            const int64_t j = d.src_beg + ((int64_t)(item % nb) << 6) + lane;
            o = d.rorg[j];
            q = d.rec[j];
            qb = BICYCLE ? d.rec2[j] : make_float2(0.f, 0.f);
        };
        int item = claim(), cur_set = -1;
        float4 q = make_float4(0.f, 0.f, 1.f, 0.f);
        float2 o = make_float2(0.f, 0.f), qb = make_float2(0.f, 0.f);
        if (item < items) fetch(item, q, o, qb);
        Recv r[RPW];
        PreciseRegs pr;
        bool first = true;
        while (item < items) {
            const int s = item / nb;
            if (s != cur_set) {                                   // (uniform) the receivers of this set: scene coordinates and precise records
                const int64_t j0 = g0 + (int64_t)s * RPW;
#pragma unroll
                for (int u = 0; u < RPW; u++) {
                    const int64_t j = j0 + u < d.hi ? j0 + u : d.hi - 1;   // clamp: results of the duplicates are not used
                    const float4 qr = d.rec[j];
                    const float2 orr = d.rorg[j];
                    pr.rx[u] = qr.x, pr.ry[u] = qr.y, pr.rox[u] = orr.x, pr.roy[u] = orr.y;
                    r[u].x = qr.x + orr.x, r[u].y = qr.y + orr.y, r[u].c = qr.z, r[u].s = qr.w;
                    asm volatile("" : "+v"(r[u].x), "+v"(r[u].y), "+v"(r[u].c), "+v"(r[u].s));  // stay in VGPRs
                }
                cur_set = s;
            }
            const int nxt = claim();
            float4 qn = q;
            float2 on = o, qbn = qb;
            if (nxt < items) fetch(nxt, qn, on, qbn);
            if (first && wave == 1) stamp(5);
            first = false;
            float ax[RPW], ay[RPW];
#pragma unroll
            for (int u = 0; u < RPW; u++) ax[u] = ay[u] = 0.0f;
            pr.sx = q.x, pr.sy = q.y, pr.sox = o.x, pr.soy = o.y;
            const float4 qs = make_float4(q.x + o.x, q.y + o.y, q.z, q.w);   // scene coordinates
            plain_pair_eval<FIELD, P2R, true>(d, k, d.p.hfov, r, g0 + (int64_t)s * RPW, qs, qb, (int32_t)(d.src_beg + ((int64_t)(item % nb) << 6) + lane), ax, ay, &pr);
            int idx;
            const float z = reduce8(lane, ax, ay, idx);
            if ((lane & 7) == 0) psum[item][idx] = z;
            item = nxt;
            q = qn;
            o = on;
            qb = qbn;
        }
        // (what this wave left in memory for wave 0 - status bits, hand-over entries - has landed before the barrier)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wave == 1) stamp(6);
        if (wave == MID_WAVES - 1) stamp(7);
    }
    __syncthreads();
    if (wave != 0) return;
    stamp(2);
    if (lane >= G) return;
    double rx = 0.0, ry = 0.0;
    {
        const int s = lane / RPW, u = lane % RPW;
        for (int b = 0; b < nb; b++) {
            rx += (double)psum[s * nb + b][2 * u];
            ry += (double)psum[s * nb + b][2 * u + 1];
        }
    }
    agent_body<MODEL, false, false, 2>(d, PH_COMBINE | PH_INTEGRATE, a, nullptr, ka_lines, rx, ry);
    if (tr != nullptr) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (lane == 0) tr[3] = wall_clock64();
    }
}

void launch_mid_tick(const Dev &d, hipStream_t st, hipEvent_t t0, hipEvent_t t1) {
    if (d.hi <= d.lo || d.mid_group < RPW || d.mid_group > MID_GROUP_MAX || d.mid_group % RPW != 0) return;
    if ((d.mid_group / RPW) * ((d.n_src - d.src_beg) >> 6) > MID_ITEMS_MAX) return;   // (csf_engine.hip: mid_fused_ok asks before)
    const dim3 g((unsigned)((d.hi - d.lo + d.mid_group - 1) / d.mid_group));
    const bool p2r = d.p.priority_rule == CSF_P2R;
#define CSF_MID(MODEL)                                                                                            \
    if (p2r) hipExtLaunchKernelGGL((mid_tick_kernel<MODEL, true>), g, dim3(mid_waves(MODEL) * WAVE), 0, st, t0, t1, 0, d);       \
    else hipExtLaunchKernelGGL((mid_tick_kernel<MODEL, false>), g, dim3(mid_waves(MODEL) * WAVE), 0, st, t0, t1, 0, d)
    switch (d.p.model) {
    case CSF_BICYCLE: CSF_MID(CSF_BICYCLE); break;
    case CSF_TWOD: CSF_MID(CSF_TWOD); break;
    case CSF_INVPEND: CSF_MID(CSF_INVPEND); break;
    case CSF_PLANARBIKE: CSF_MID(CSF_PLANARBIKE); break;
    default: CSF_MID(CSF_PLANARPOINT); break;
    }
#undef CSF_MID
}

}  // namespace csf
