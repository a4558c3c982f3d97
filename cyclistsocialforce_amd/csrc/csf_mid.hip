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
//   the others    ... while they form the group's repulsive sums, item by item from an LDS counter, sources in the lanes, straight
//                 from the records in memory, asked for one item ahead.  TwoD field: an item is ONE receiver (wave-uniform, in
//                 scalar registers) against four batches of 64 sources, cull first as in csf_pair.hip - two batches at a time
//                 through the field-of-view test on packed arithmetic, the tracked sources to a queue in LDS, the packed field on
//                 what the queue holds, the rare pairs (near, marginal, on the line ahead of a source) one per lane on the precise
//                 records.  Bicycle field: an item is four receivers against one batch (csf_pair_dev.h: plain_pair_eval - the very
//                 code of pair_kernel).  Column sum by lane exchange, one partial per item (and receiver) to LDS;
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
// else eight (InvPendulum; the BalancingRider's per-agent chain is longer still and its one-launch tick was measured no faster than
// two launches - 16.4 against 15.6 us at 1 024 riders -: csf_engine.hip keeps that class on two launches) - one workgroup per CU either way, and one more wave per SIMD on the pair sums
__host__ __device__ constexpr int mid_waves(int model) { return model == CSF_INVPEND ? 8 : 12; }
constexpr int MID_GROUP_MAX = 32;             // road users (slots) of a group at most
constexpr int MID_ITEMS_MAX = 384;            // (receiver set, source batch) items of a group at most: 12 KB of partial sums
constexpr int MID_SB = 4;                     // TwoD field: source batches of an item (one receiver against 256 sources)

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
    // TwoD field (cull first, below): the sources of the item a wave is working on, scene coordinates, SoA; its queue of kept ones
    __shared__ float tx[BICYCLE ? 1 : MID_WAVES - 1][BICYCLE ? 1 : MID_SB * WAVE], ty[BICYCLE ? 1 : MID_WAVES - 1][BICYCLE ? 1 : MID_SB * WAVE];
    __shared__ float tc[BICYCLE ? 1 : MID_WAVES - 1][BICYCLE ? 1 : MID_SB * WAVE], ts[BICYCLE ? 1 : MID_WAVES - 1][BICYCLE ? 1 : MID_SB * WAVE];
    __shared__ unsigned short kq[BICYCLE ? 1 : MID_WAVES - 1][BICYCLE ? 1 : MID_SB * WAVE];
    __shared__ float4 grec[BICYCLE ? 1 : MID_GROUP_MAX];         // ... and the group's receivers: precise records and their origins
    __shared__ float2 gorg[BICYCLE ? 1 : MID_GROUP_MAX];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = d.mid_group, sets = G / RPW;
    const int nb = (int)((d.n_src - d.src_beg) >> 6);
    const int ngr = (nb + MID_SB - 1) / MID_SB;               // TwoD field: groups of MID_SB source batches
    const int items = BICYCLE ? sets * nb : G * ngr;
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
    if (!BICYCLE && wave == 1 && lane < G) {                      // (one round trip for the whole group, not one per item)
        const int64_t j = g0 + lane < d.hi ? g0 + lane : d.hi - 1;
        grec[lane] = d.rec[j];
        gorg[lane] = d.rorg[j];
    }
    __syncthreads();
    if (wave == 0) {
        stamp(0);
        if (lane < G) agent_body<MODEL, false, false, 1>(d, PH_DEST, a, nullptr, ka_lines, 0.0, 0.0);
        stamp(1);
    } else if (!BICYCLE) {
        // The pair sums under the TwoD field, CULL FIRST (as csf_pair.hip does it for large populations): an item is ONE receiver
        // - wave-uniform, in scalar registers - against MID_SB batches of 64 sources, one record per lane and batch straight from
        // memory (asked for one item ahead).  Every lane tests its sources (intersection.py:690-745: ~12 instructions), the wave
        // ballots, and the tracked ones - a third under the default field of view - are appended to the wave's queue in LDS; the
        // field (vehicle.py:1560-1648: ~100 instructions for two sources per lane on packed arithmetic) runs on what the queue
        // holds.  The rare pairs - closer than rnear, within rounding of a field-of-view edge or of the line ahead of the source -
        // never enter the queue: they are decided and evaluated one per lane on the precise records, as plain_pair_eval does
        // it.  (The plain evaluation - mask after the field, one receiver at a time, four per item - took 520 instructions per
        // 256 pairs; this takes ~230, and the pair sums were the longer of the two legs in front of the barrier.)
        if (wave == 1) stamp(4);
        const int w1 = wave - 1;
        PairConsts k = d.pc;
        asm volatile("" : "+v"(k.sg0), "+v"(k.sg1), "+v"(k.sg2), "+v"(k.sg3), "+v"(k.e0), "+v"(k.e1), "+v"(k.lf0), "+v"(k.kexp), "+v"(k.chs));
        auto claim = [&]() {
            int got = 0;
            if (lane == 0) got = atomicAdd(&next_item, 1);
            return __builtin_amdgcn_readfirstlane(got);
        };
        auto fetch = [&](int item, float4 (&q)[MID_SB], float2 (&o)[MID_SB]) {
            const int b0 = (item % ngr) * MID_SB;
#pragma unroll
            for (int b = 0; b < MID_SB; b++) {
                const int bb = b0 + b < nb ? b0 + b : nb - 1;           // (a short last group: the duplicates are not looked at)
                const int64_t j = d.src_beg + ((int64_t)bb << 6) + lane;
                o[b] = d.rorg[j];
                q[b] = d.rec[j];
            }
        };
        int item = claim();
        float4 q[MID_SB];
        float2 o[MID_SB];
#pragma unroll
        for (int b = 0; b < MID_SB; b++) q[b] = make_float4(0.f, 0.f, 1.f, 0.f), o[b] = make_float2(0.f, 0.f);
        if (item < items) fetch(item, q, o);
        bool first = true;
        while (item < items) {
            const int g = item / ngr, b0 = (item % ngr) * MID_SB;
            const int nbi = nb - b0 < MID_SB ? nb - b0 : MID_SB;       // batches of this item
            const int64_t jr = g0 + g;
            const bool real = jr < d.hi;                               // (a slot behind the last road user: nothing to sum)
            // the receiver: wave-uniform (scalar registers), scene coordinates for the fast path, the precise record beside it
            const float4 qr = grec[g];
            const float2 orr = gorg[g];
            Recv ru;
            ru.x = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(qr.x + orr.x)));
            ru.y = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(qr.y + orr.y)));
            ru.c = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(qr.z)));
            ru.s = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(qr.w)));
            // this item's sources: kept (registers), and as scene coordinates in the wave's tile
            float4 qs[MID_SB];
            float2 os[MID_SB];
#pragma unroll
            for (int b = 0; b < MID_SB; b++) {
                qs[b] = q[b], os[b] = o[b];
                tx[w1][(b << 6) + lane] = q[b].x + o[b].x;
                ty[w1][(b << 6) + lane] = q[b].y + o[b].y;
                tc[w1][(b << 6) + lane] = q[b].z;
                ts[w1][(b << 6) + lane] = q[b].w;
            }
            const int nxt = claim();
            if (nxt < items) fetch(nxt, q, o);
            if (first && wave == 1) stamp(5);
            first = false;
            float ax = 0.0f, ay = 0.0f;
            int qlen = 0;                                             // (wave-uniform) kept sources waiting in kq
            // the packed field on the first 128 queued sources, or on whatever is left (the queue is then empty)
            auto pass = [&](auto full) {
                constexpr bool FULL = decltype(full)::value;
                const int n = FULL ? CHUNK : qlen;
                const bool v0 = lane < n, v1 = lane + WAVE < n;
                int i0 = kq[w1][lane], i1 = kq[w1][WAVE + lane];
                if (!FULL) i0 = v0 ? i0 : 0, i1 = v1 ? i1 : 0;
                field_twod_x2<FULL, false>(k, ru, lds_pair_b(tx[w1], i0, i1), lds_pair_b(ty[w1], i0, i1), lds_pair_b(tc[w1], i0, i1),
                                           lds_pair_b(ts[w1], i0, i1), v0, v1, ax, ay);
                if (FULL) {                                           // what is queued behind the 128 moves up (at most 127 entries:
                    const int rest = qlen - CHUNK;                    //  two batches are appended at a time)
                    const unsigned short mv = kq[w1][CHUNK + lane], mw = kq[w1][(CHUNK + WAVE + lane) & (MID_SB * WAVE - 1)];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (lane < rest) kq[w1][lane] = mv;
                    if (lane + WAVE < rest) kq[w1][WAVE + lane] = mw;
                    qlen = rest;
                } else {
                    qlen = 0;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            };
            if (real) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // the rare pairs of batch b - near, marginal, or on the line ahead of the source: (dx, dy) from the precise records (the
                // very expression of precise_delta) and the decisions on them (csf_pair_dev.h: plain_pair_eval)
                auto rare = [&](const int b, const bool fix) {
                    const float4 qq = make_float4(qs[b].x + os[b].x, qs[b].y + os[b].y, qs[b].z, qs[b].w);   // (the tile's very values)
                    const int32_t as = (int32_t)(d.src_beg + ((int64_t)(b0 + b) << 6) + lane);
                    const float px = (qr.x - qs[b].x) + (orr.x - os[b].x), py = (qr.y - qs[b].y) + (orr.y - os[b].y);
                    float r2p = px * px + py * py;
                    bool edge;
                    const bool other = fix & (as != (int32_t)jr);
                    const bool inp = tracked_precise<P2R>(d.pc, k.chs, ru, px, py, r2p, edge) & other;
                    edge = edge & other;
                    const bool side = other & side_undecided(d.pc, qq, px, py, r2p) & (d.edge != nullptr);   // (nobody to hand it to: the pair's own sign)
                    r2p = fmaxf(r2p, 1e-30f);
                    float F, hx, hy;
                    field_twod(k, ru, qq, px, py, r2p, F, hx, hy, side ? 1.0f : 0.0f);
                    if (d.edge != nullptr && (ballot1(edge) | ballot1(side)) != 0ull) {   // undecidable even so: the per-agent phase decides
                        if (edge | (side & inp)) {
                            float F2 = 0.0f, h2x = 0.0f, h2y = 0.0f;
                            if (side) field_twod(k, ru, qq, px, py, r2p, F2, h2x, h2y, -1.0f);
                            edge_handover(d, (int32_t)jr, as, d.p.hfov, F * hx, F * hy, inp, side, F2 * h2x, F2 * h2y);
                        }
                    }
                    F = inp ? F : 0.0f;
                    ax += F * hx;
                    ay += F * hy;
                };
#pragma unroll
                for (int b = 0; b < MID_SB; b += 2) {
                    if (b >= nbi) break;                              // (uniform)
                    const bool two = b + 1 < nbi;
                    // two batches per lane on packed arithmetic: the test of tracked_m (csf_field.h) with the band of its rounding
                    const v2f sx{qs[b].x + os[b].x, qs[b + 1].x + os[b + 1].x}, sy{qs[b].y + os[b].y, qs[b + 1].y + os[b + 1].y};
                    const v2f sc{qs[b].z, qs[b + 1].z}, ss{qs[b].w, qs[b + 1].w};
                    const v2f dx = ru.x - sx, dy = ru.y - sy;          // vehicle.py:1615-1616
                    const v2f r2 = dx * dx + dy * dy;
                    const v2f t = -(dx * ru.c + dy * ru.s);
                    const v2f gg = t * fabs2(t) + k.chs * r2;
                    const v2f band = d.pc.fovA * r2 + d.pc.fovB;
                    const v2f gb = gg + band;
                    const v2f cross = fabs2(dy * sc - dx * ss);
                    const v2f Ts = d.pc.fovT1 + d.pc.fovT0 * (0.0625f * r2 + 4.0f);     // (rho <= r2 / 16 + 4)
                    const unsigned long long both = 0ull - (unsigned long long)two;
                    unsigned long long in0 = ballot1(gg.x > 0.0f), in1 = ballot1(gg.y > 0.0f);
                    unsigned long long kp0 = ballot1(gb.x > 0.0f), kp1 = ballot1(gb.y > 0.0f);
                    unsigned long long lt0 = ballot1(gg.x < band.x), lt1 = ballot1(gg.y < band.y);
                    if (P2R) {
                        const v2f side = ru.s * dx - ru.c * dy, sb = d.pc.sideA * r2 + d.pc.sideB;
                        in0 &= ~ballot1(side.x > 0.0f), in1 &= ~ballot1(side.y > 0.0f);
                        kp0 &= ~ballot1(side.x > sb.x), kp1 &= ~ballot1(side.y > sb.y);
                        lt0 |= ballot1(side.x > -sb.x), lt1 |= ballot1(side.y > -sb.y);
                    }
                    const unsigned long long fx0 = ballot1(r2.x < d.pc.rnear2) | ballot1(cross.x < Ts.x) | (kp0 & lt0);
                    const unsigned long long fx1 = (ballot1(r2.y < d.pc.rnear2) | ballot1(cross.y < Ts.y) | (kp1 & lt1)) & both;
                    const unsigned long long m0 = in0 & ~fx0, m1 = in1 & ~fx1 & both;
                    const int n0 = __builtin_popcountll(m0);
                    if ((m0 >> lane) & 1ull) {
                        const int at = __builtin_amdgcn_mbcnt_hi((unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m0, (unsigned)qlen));
                        kq[w1][at] = (unsigned short)(4 * ((b << 6) + lane));
                    }
                    if ((m1 >> lane) & 1ull) {
                        const int at = __builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, (unsigned)(qlen + n0)));
                        kq[w1][at] = (unsigned short)(4 * (((b + 1) << 6) + lane));
                    }
                    qlen = __builtin_amdgcn_readfirstlane(qlen + n0 + __builtin_popcountll(m1));
                    if (__builtin_expect(fx0 != 0ull, 0)) rare(b, ((fx0 >> lane) & 1ull) != 0ull);
                    if (__builtin_expect(fx1 != 0ull, 0)) rare(b + 1, ((fx1 >> lane) & 1ull) != 0ull);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (qlen >= CHUNK) pass(std::true_type{});        // (at most 127 + 128 were queued: one pass leaves fewer than 128)
                }
                if (qlen > 0) pass(std::false_type{});
            }
            // column sum: x in the lower half of the wave, y in the upper, then within the halves (csf_pair.hip)
            float v = swap_add32(ax, ay);
            v += dpp<DPP_ROW_ROR8>(v);
            v += dpp<DPP_XOR1>(v);
            v += dpp<DPP_XOR2>(v);
            v += dpp<DPP_HALF_MIRROR>(v);
            const auto rr = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
            v = __uint_as_float(rr[0]) + __uint_as_float(rr[1]);
            if ((lane & 31) == 0) (&psum[0][0])[2 * item + (lane >> 5)] = v;
            item = nxt;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wave == 1) stamp(6);
        if (wave == MID_WAVES - 1) stamp(7);
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
    if (BICYCLE) {
        const int s = lane / RPW, u = lane % RPW;
        for (int b = 0; b < nb; b++) {
            rx += (double)psum[s * nb + b][2 * u];
            ry += (double)psum[s * nb + b][2 * u + 1];
        }
    } else {
        for (int b = 0; b < ngr; b++) {
            rx += (double)(&psum[0][0])[2 * (lane * ngr + b)];
            ry += (double)(&psum[0][0])[2 * (lane * ngr + b) + 1];
        }
    }
    agent_body<MODEL, false, false, 2>(d, PH_COMBINE | PH_INTEGRATE, a, nullptr, ka_lines, rx, ry);
    if (tr != nullptr) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (lane == 0) tr[3] = wall_clock64();
    }
}

// false: nothing was launched (a shape this kernel is not built for) - the caller must not count the tick as taken
bool launch_mid_tick(const Dev &d, hipStream_t st, hipEvent_t t0, hipEvent_t t1) {
    if (d.hi <= d.lo || d.mid_group < RPW || d.mid_group > MID_GROUP_MAX || d.mid_group % RPW != 0) return false;
    if ((d.mid_group / RPW) * ((d.n_src - d.src_beg) >> 6) > MID_ITEMS_MAX) return false;   // (csf_engine.hip: mid_fused_ok asks before)
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
    return true;
}

}  // namespace csf
