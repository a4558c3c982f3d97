// csf_pair.hip — all-pairs repulsive force field + field-of-view mask + column sum (gfx950).
//
// Replaces, per tick:  SocialForceIntersection.get_untracked_foes (intersection.py:690-745),
// the N calls of vehicle.calcRepulsiveForce (intersection.py:814-823; vehicle.py:1560-1648 or
// 1054-1147) and the column sum of intersection.py:841-843;  RoadEdge.calcRepulsiveForce
// (intersection.py:226-242) for the static-obstacle term.
//
// Mapping (CDNA4, 64-wide waves): SOURCES sit in the lanes, RECEIVERS are wave-uniform.  A workgroup owns 16 or 32
// receivers and streams the fp32 source records (x, y, cos psi, sin psi) of one chunk through an LDS tile that its four
// waves share; the waves take the receivers one at a time (an LDS counter), each lane accumulates its sources'
// contribution, and gfx950 lane exchanges (v_permlane32/16_swap, DPP adds) form the column sum of a receiver, which is
// added to an LDS accumulator in a fixed order.  No atomics on the sums: the result is bit-reproducible.
//
// The kernel is VALU-issue bound (tools/valu_ubench.hip: ~3-4.5 cycles per wave64 fp32 instruction, ~8.3 per
// transcendental, v_pk_* at ~4.7 for two results), so the work is organised to issue as little as possible:
//   1. CULL FIRST.  The field-of-view test costs ~12 instructions, the field itself ~70.  With the
//      reference's default 120 degree field of view two thirds of all pairs are masked.  Every lane tests its
//      source, the wave ballots, and the lanes that pass append their tile index to a per-receiver queue in
//      LDS (prefix = v_mbcnt of the ballot).  The field is evaluated only on full batches popped from the
//      queue, so all 64 lanes do useful work there.
//   2. TWO PAIRS PER LANE in the field evaluation, written on float2 so that hipcc emits v_pk_fma_f32 /
//      v_pk_mul_f32 / v_pk_add_f32.
//   3. Wave-uniform operands are kept in VGPRs (an SGPR operand makes a VALU instruction slower here).
//   4. WHOLE BATCHES of 64 spatially binned records are classified first from their bounding circle (csf_bin.hip):
//      outside the field of view or beyond the far-field radius -> skipped, inside -> evaluated without the test.
// The math is trig-free: every angle of the reference enters only through sin/cos, which are dot and cross
// products of unit vectors here.
#include "csf_pair_dev.h"

namespace csf {


// ---- culling kernel (TwoD field): classify batches -> test -> ballot -> LDS queue -> packed field ---------
// CLASSIFY: the records are streamed in spatially binned order (csf_bin.hip) and every batch of 64 carries a
// bounding circle.  Per receiver and tile, lane b classifies batch b against the field-of-view cone:
//   outside (smallest bearing in the circle > hfov/2)  -> skipped without touching its records,
//   inside  (largest bearing in the circle  < hfov/2)  -> all 64 lanes queued without per-lane tests,
//   else                                               -> exact per-lane test (intersection.py:690-745).
// Both shortcuts keep a 1e-4 margin in the cosine and need the receiver outside the circle, so the exact test
// decides every borderline source: these two leave the results identical with and without CLASSIFY.
// A batch is also skipped when all of it lies beyond the far-field radius k.rfar (csf_engine.hip: far_radius),
// where the contributions are below the resolution of the fp32 column sum.
// BINR (large populations): the RECEIVERS of a workgroup are consecutive places of the binned order as well, i.e.
// neighbours in space, so a whole tile of 1024 sources that lies beyond the far-field radius of the group's bounding
// circle is skipped before it is loaded (at 262 144 agents in 800 m: 4 of 5 tiles).  Tile and receivers are then held
// relative to the ORIGIN OF THE GROUP (that of the batch of its first receiver's place, csf_dev.h: recb, borg): a source that matters is
// near the group, so the fp32 difference receiver - source keeps 2^-24 of (pair distance + group extent) at any extent
// of the scene - 6e-5 m would be the resolution of scene coordinates in the 1 600 m of config 5.  (Smaller populations take their
// receivers in slot order: a workgroup then holds a random sample of receivers and every workgroup carries the same
// load - with neighbours the few workgroups whose tile lies in front of ALL their receivers end the kernel 10 % later.)
// DYN (small receiver blocks, i.e. shards of a small population: launch_pair): the 16 receivers of the workgroup are
// handed to its waves one at a time through an LDS counter instead of four per wave.  With few workgroups the kernel
// ends with its longest single wave (a lone wave issues one dependent instruction every ~8 cycles); sharing the
// receivers shortens that wave.  At full size the SIMDs are issue-bound either way and the variant is not used.
// RPB: receivers of a workgroup, 16, or 32 where the grid stays large enough (DYN only: with dynamic hand-out a
// workgroup of twice the receivers shares one tile fill and one start-up between them: 146 -> 141 us at N = 16 384)
// REACH (with the far-field cull on): every candidate batch goes through the packed reach test keep_x2, two batches at a
// time, and only the sources it keeps are queued for the field.
// CW: waves of a workgroup, 4, or 8 (DYN, RPB 32: launch_cull_dyn): a workgroup of 8 waves holds a
// tile of 2048 sources (32 batches), so a visit - one receiver against one tile - covers twice the sources: half the visits
// (each with its claim, record, column sum), and twice the sources to evaluate per visit, i.e. fuller evaluation passes.
// SEG (several parameter sets in the class-segmented order, csf_engine.hip: rebin): ONE grid for the runs of all sets whose
// field is the TwoD one - blockIdx.y counts the source chunks of all runs, and a workgroup takes its run's constants, source
// range and partial-sum slots from Dev::segtab (a launch per run left the device idle while each run's last workgroups ended).
template <bool P2R, bool CLASSIFY, bool BINR, bool DYN, int RPB = WPB * RPW, bool REACH = false, int CW = WPB, bool SEG = false>
__global__ __launch_bounds__(CW * WAVE, CSF_CULL_WAVES) void pair_cull_kernel(const Dev dk) {
    static_assert(!SEG || (DYN && CLASSIFY && !BINR), "the segmented grid is built into the classified variants with receivers in slot order");
    Dev dseg;
    int by = (int)blockIdx.y;
    double seg_hfov = 0.0;
    if (SEG) {
        const int ln = threadIdx.x & (WAVE - 1);
        const int fb = dk.segtab[ln < dk.n_seg ? ln : 0].first_by;
        const int sg = __builtin_popcountll(ballot1(ln < dk.n_seg && (int)blockIdx.y >= fb)) - 1;
        const SegDev &t = dk.segtab[sg];
        dseg = dk;
        by = (int)blockIdx.y - t.first_by;
        dseg.src_beg = t.src_beg;
        dseg.n_src = t.n_src;
        dseg.chunk_units = t.chunk_units;
        dseg.part_base = t.part_base;
        PairConsts pc = t.pc;                 // the set's field, field of view, far-field radius and reach test ...
        pc.fovA = dk.pc.fovA, pc.fovB = dk.pc.fovB, pc.sideA = dk.pc.sideA, pc.sideB = dk.pc.sideB;   // ... the rounding bands of this tick
        pc.fovT0 = dk.pc.fovT0, pc.fovT1 = dk.pc.fovT1, pc.clsk = dk.pc.clsk;
        pc.fovP1 = dk.pc.fovP1, pc.fovP2 = dk.pc.fovP2, pc.sideP0 = dk.pc.sideP0, pc.sideP1 = dk.pc.sideP1;
        dseg.pc = pc;
        seg_hfov = t.hfov;
    }
    const Dev &d = SEG ? (const Dev &)dseg : dk;
    static_assert(RPB == WPB * RPW || (DYN && (RPB % (WPB * RPW) == 0 || RPB == 2 * RPW) && RPB <= WAVE), "other workgroup sizes need the dynamic hand-out");
    static_assert(CW == WPB || (CW == 2 * WPB && DYN && CLASSIFY && (RPB == 32 || RPB == 16)), "the wide workgroup is built into the variants with 16 / 32 receivers");
    constexpr int BLOCKW = CW * WAVE;             // threads of a workgroup
    constexpr int TL = TILE2 / WPB * CW;          // sources of a tile
    constexpr int NBT = TL / WAVE;                // batches of a tile: 16 or 32
    constexpr int RPP = WAVE / NBT;               // receivers one classification pass of a wave covers: 4 or 2
    constexpr unsigned NBMASK = NBT >= 32 ? 0xFFFFFFFFu : ((1u << (NBT & 31)) - 1u);
    static_assert(!REACH || (CLASSIFY && DYN), "the reach test is built into the classified, dynamically handed-out variant");
    __shared__ float tx[TL], ty[TL], tc[TL], ts[TL];              // SoA: the two records of a lane load straight
    __shared__ float4 tbnd[NBT];                                   // into the halves of a packed register pair
    __shared__ unsigned short queue[CW][QCAP];  // one queue per wave, drained after each receiver; holds BYTE offsets
                                                 // into the tile arrays (4 x index <= 4092)
    __shared__ float4 rrec[RPB];
    __shared__ int ragent[RPB];              // slot of every receiver of the workgroup (-1: none)
    __shared__ unsigned bmask[DYN ? RPB : 1];   // DYN: candidate (| inside << 16, 16 batches) batch masks of every receiver
    __shared__ unsigned bmask_in[DYN && NBT > 16 ? RPB : 1];   // 32 batches: the inside masks
    constexpr int NCAP = 32;                           // near pairs noted by one wave and tile: receiver << 16 | tile index (32, not
                                                       // 64: with 256 bytes less a workgroup fits 20 KB of LDS, eight to a CU)
    __shared__ unsigned nlist[CW][NCAP];
    __shared__ float racc[2][DYN ? RPB : 1];    // DYN: column sums of the workgroup's receivers
    __shared__ int next_recv;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t j0 = d.lo + ((int64_t)blockIdx.x * CW + wave) * RPW;
    int64_t ibeg, iend;
    source_chunk(d, ibeg, iend, by);
    const uint64_t t_start = d.trace ? wall_clock64() : 0;
    if (cold_args().chase_clock != nullptr && lane == 0 && wave == 0)      // (measurement aid: when did this launch's first wave start?)
        atomicMin(cold_args().chase_clock + 8 * cold_args().chase_slot + 0, (unsigned long long)wall_clock64());
    float2 og = make_float2(0.f, 0.f);   // BINR: origin of the workgroup (uniform: scalar loads), else the scene's
    if (BINR) {
        const int64_t jg = d.lo + (int64_t)blockIdx.x * RPB;
        og = d.borg[(d.rlist ? (int64_t)d.rlist[jg - d.lo] : jg) >> 6];
    }

    // the workgroups of the first source chunk also emit the bounding circles of the next tick (one wave per
    // batch): a separate launch per tick would cost more in launch gaps than in work
    if (CLASSIFY && d.bnd_next != nullptr && by == 0) {
        for (int64_t b = (d.src_beg >> 6) + (int64_t)blockIdx.x * CW + wave; b * WAVE < d.n_src; b += (int64_t)gridDim.x * CW)
            batch_circle(d, b, lane, d.bnd_margin, d.bnd_next, d.part4 != nullptr);
    }

    float ax[RPW], ay[RPW];     // the receivers themselves stay in LDS (rrec); one at a time is held in registers
    int qhead = 0, qlen = 0;    // wave-uniform ring state of the queue
    unsigned evals = 0;         // pair evaluations of this wave (wave-uniform: scalar adds; written out for csf_count_pairs)
    unsigned tests = 0;         // ... sources put through the per-lane tests, and full | partial << 16 evaluation passes
    unsigned pops = 0;
    Recv ru{0.f, 0.f, 1.f, 0.f};  // the receiver being worked on (wave-uniform, kept in VGPRs)
#pragma unroll
    for (int u = 0; u < RPW; u++) ax[u] = ay[u] = 0.0f;
    PairConsts k = d.pc;
    asm volatile("" : "+v"(k.sg0), "+v"(k.sg1), "+v"(k.sg2), "+v"(k.sg3), "+v"(k.e0), "+v"(k.e1), "+v"(k.lf0),
                 "+v"(k.kexp), "+v"(k.chs));
    if (REACH) asm volatile("" : "+v"(k.tA0), "+v"(k.tA1), "+v"(k.tB0), "+v"(k.tB1));

    // Near pairs (precise_delta): what the field evaluation meets closer than k.rnear is noted here - receiver << 16 |
    // tile index - and corrected once per wave and tile (near_drain, in the tile loop).
    int nlen = 0, cur_recv = 0;         // (wave-uniform) entries of the wave's list; the receiver being worked on
    unsigned ndrop = 0;                 // ... entries that found it full
    auto near_note = [&](bool nr, int tile_index) {
        const unsigned long long m = ballot1(nr);
        if (nr) {
            const int at = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)nlen));
            if (at < NCAP) nlist[wave][at] = ((unsigned)cur_recv << 16) | (unsigned)tile_index;   // (full: left uncorrected)
        }
        const int nn = nlen + __builtin_popcountll(m);
        ndrop += nn > NCAP ? (unsigned)(nn - NCAP) : 0u;      // (reported once, at the end: csf_near_dropped - the tests assert 0)
        nlen = __builtin_amdgcn_readfirstlane(nn < NCAP ? nn : NCAP);
    };
    // pop CHUNK (or, when draining, whatever is left) queued sources of receiver u; the field takes two per lane
    auto pop = [&](int u, auto full) {
        constexpr bool FULL = decltype(full)::value;
        const int n = FULL ? CHUNK : (qlen < CHUNK ? qlen : CHUNK);
        int i0 = queue[wave][(qhead + lane) & (QCAP - 1)];
        int i1 = queue[wave][(qhead + WAVE + lane) & (QCAP - 1)];
        const bool v0 = lane < n, v1 = lane + WAVE < n;
        if (!FULL) i0 = v0 ? i0 : 0, i1 = v1 ? i1 : 0;
        unsigned long long n0, n1;
        field_twod_x2<FULL, true>(k, ru, lds_pair_b(tx, i0, i1), lds_pair_b(ty, i0, i1), lds_pair_b(tc, i0, i1),
                                  lds_pair_b(ts, i0, i1), v0, v1, ax[u], ay[u], &n0, &n1);
        if (__builtin_expect((n0 | n1) != 0ull, 0)) {   // rare: a pair among the 128 to be corrected (the queue still holds their offsets)
            near_note(((n0 >> lane) & 1ull) != 0ull, (int)queue[wave][(qhead + lane) & (QCAP - 1)] >> 2);
            near_note(((n1 >> lane) & 1ull) != 0ull, (int)queue[wave][(qhead + WAVE + lane) & (QCAP - 1)] >> 2);
        }
        qhead = __builtin_amdgcn_readfirstlane((qhead + n) & (QCAP - 1));
        qlen = __builtin_amdgcn_readfirstlane(qlen - n);
        if (FULL) pops += 1u;             // (the evaluations of the full passes are added at the end: 128 each)
        else evals += (unsigned)n, pops += 0x10000u;
    };
    // the last (at most 64) queued sources of a receiver: one per lane through the unpacked field - about 60 % of the
    // instructions of a packed evaluation whose second half would be empty
    auto pop_tail = [&](int u) {
        const bool v = lane < qlen;
        const int o = v ? (int)queue[wave][(qhead + lane) & (QCAP - 1)] : 0;
        const float4 q = make_float4(*(const float *)((const char *)tx + o), *(const float *)((const char *)ty + o),
                                     *(const float *)((const char *)tc + o), *(const float *)((const char *)ts + o));
        const float dx = ru.x - q.x, dy = ru.y - q.y, r2 = dx * dx + dy * dy;
        float F, gx, gy;
        field_twod<false>(k, ru, q, dx, dy, fmaxf(r2, 1e-30f), F, gx, gy);
        {   // (the wave mask of ONE compare, masked as a scalar: ballot1)
            // (+ np.sign(phi) within rounding of phi = 0: csf_field.h field_twod_x2)
            const float inv1 = fast_rsq(fmaxf(r2, 1e-30f));
            const unsigned long long nm = (ballot1(r2 < k.rnear2) | ballot1(__builtin_fabsf((dy * q.z - dx * q.w) * inv1) <= k.fovT0 + k.fovT1)) &
                                          (qlen >= WAVE ? ~0ull : ((1ull << qlen) - 1ull));
            if (__builtin_expect(nm != 0ull, 0)) near_note(((nm >> lane) & 1ull) != 0ull, o >> 2);
        }
        F = v ? F : 0.0f;
        ax[u] += F * gx;
        ay[u] += F * gy;
        evals += (unsigned)qlen;
        pops += 0x10000u;
        qhead = __builtin_amdgcn_readfirstlane((qhead + qlen) & (QCAP - 1));
        qlen = 0;
    };

    // a tile: 1024 places of the binned order in scene coordinates (recs: the copy the per-agent kernel maintains beside
    // the precise records; without it - unbinned - offset + origin of the slot)
    auto fill_tile = [&](int64_t base, int cnt, int tid) {
        for (int t = tid; t < cnt; t += BLOCKW) {
            float4 q;
            if (BINR) {          // relative to the group's origin: (relative to its batch's) + (batch origin - group origin), the bracket exact
                const float2 bo = d.borg[__builtin_amdgcn_readfirstlane((int)((base + t) >> 6))];   // (one batch per wave and round: a scalar load)
                q = d.recb[base + t];
                q.x += bo.x - og.x, q.y += bo.y - og.y;
            } else if (d.recs_valid) {
                q = d.recs[base + t];
            } else {
                const int32_t a = d.perm[base + t];
                const float2 o = d.rorg[a];
                q = d.rec[a];
                q.x += o.x, q.y += o.y;
            }
            tx[t] = q.x;
            ty[t] = q.y;
            tc[t] = q.z;
            ts[t] = q.w;
        }
        if (CLASSIFY && tid < (cnt >> 6)) {
            float4 bb = d.bnd[(base >> 6) + tid];             // (scene coordinates)
            if (BINR) bb.x -= og.x, bb.y -= og.y;
            tbnd[tid] = bb;
        }
        if (DYN && tid == BLOCKW - 1) next_recv = 0;
    };
    // BINR: the tiles this receiver group can meet before the next re-binning are listed (csf_dev.h: Dev::clist) - the workgroup
    // takes its share of the list (source chunk c of n: entries [c L / n, (c + 1) L / n)) and of the tiles of the sentinel
    // tail, which are always visited, instead of walking every tile of its chunk
    bool uselist = false;
    int li = 0, li_end = 0;
    int64_t tt = 0, ntl = 0;
    if (BINR && d.clist != nullptr && d.clist_tile == TL && d.clist_rpb == RPB) {
        const int lc = d.ccount[blockIdx.x];
        if (lc >= 0) {
            uselist = true;
            li = (int)((int64_t)lc * blockIdx.y / gridDim.y);
            li_end = (int)((int64_t)lc * (blockIdx.y + 1) / gridDim.y);
            ntl = (d.n_src - d.src_beg + TL - 1) / TL;
            tt = (int64_t)d.ctail + blockIdx.y;
            iend = d.n_src;
        }
    }
    // first tile and the workgroup's receiver records travel together: one global round trip, not two
    if (!uselist && ibeg >= iend) return;  // (uniform) nothing to do for this chunk
    if (!BINR) fill_tile(ibeg, (int)((iend - ibeg) < TL ? (iend - ibeg) : TL), (int)threadIdx.x);
    if (threadIdx.x < RPB) {
        const int64_t j = d.lo + (int64_t)blockIdx.x * RPB + threadIdx.x;
        const int64_t jc = j < d.hi ? j : d.hi - 1;          // clamp: results of the duplicates are not stored
        if (BINR) {  // receiver slot jc - lo of this rank -> place of the binned order -> slot
            const int64_t p = d.rlist ? (int64_t)d.rlist[jc - d.lo] : jc;
            const float2 bo = d.borg[p >> 6];
            float4 q = d.recb[p];
            q.x += bo.x - og.x, q.y += bo.y - og.y;                   // (the very expression of the tile fill)
            rrec[threadIdx.x] = q;
            ragent[threadIdx.x] = j < d.hi ? d.perm[p] : -1;
        } else {
            if (d.recs_valid) {                               // (uniform) scene coordinates, identical with the slot's entry of recs
                rrec[threadIdx.x] = d.recg[jc];
            } else {
                const float2 o = d.rorg[jc];
                float4 q = d.rec[jc];
                q.x += o.x, q.y += o.y;
                rrec[threadIdx.x] = q;
            }
            ragent[threadIdx.x] = j < d.hi ? (int)jc : -1;
        }
        if (DYN) racc[0][threadIdx.x] = racc[1][threadIdx.x] = 0.0f;
    }
    __syncthreads();
    float gx = 0.f, gy = 0.f, gr = 0.f;   // BINR: bounding circle of the workgroup's receivers (the same in every wave)
    if (BINR) {
        // (the places of road users that left since the last re-binning hold sentinel records, 1e15 m away: they take no part -
        // a circle that reached out to them would pass every tile, and its extent would blow up the rounding band below
        // until every pair of the group's receivers went down the exact path)
        const float4 q = rrec[lane & (RPB - 1)];
        const bool real = rec_is_real(q);
        float x0 = real ? q.x : 3e38f, x1 = real ? q.x : -3e38f, y0 = real ? q.y : 3e38f, y1 = real ? q.y : -3e38f;
#pragma unroll
        for (int o = RPB / 2; o > 0; o >>= 1) {
            x0 = fminf(x0, __shfl_xor(x0, o, WAVE));
            x1 = fmaxf(x1, __shfl_xor(x1, o, WAVE));
            y0 = fminf(y0, __shfl_xor(y0, o, WAVE));
            y1 = fmaxf(y1, __shfl_xor(y1, o, WAVE));
        }
        if (x1 < x0) x0 = x1 = y0 = y1 = 0.f;                     // nobody here: an empty circle at the group's origin
        const float w = x1 - x0, h = y1 - y0;
        gx = 0.5f * (x0 + x1), gy = 0.5f * (y0 + y1);
        gr = 0.5f * fast_sqrt(w * w + h * h) * 1.0001f + 1e-4f;   // rounded up: must contain
        // the same in every lane: kept in scalar registers (as vector registers one of them was spilled, and reloaded from
        // scratch memory in front of every tile's far test)
        asm volatile("" : "+v"(gx), "+v"(gy), "+v"(gr));   // (the finished values, not an operand of theirs, go to the scalar registers)
        gx = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(gx)));
        gy = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(gy)));
        gr = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(gr)));
        // The rounding band of the field-of-view test (csf_field.h: keep_x2) in THIS frame: a source the reach test keeps is
        // within the far-field radius of its receiver, a receiver within the group's circle around the group's origin - no
        // coordinate the test sees exceeds |centre| + radius + far-field radius (csf_engine.hip: set_fov_band has the scene's
        // extent for receivers in slot order).
        if (REACH) k.fovT1 = k.fovT0 * 0.38f * (fabsf(gx) + fabsf(gy) + 2.0f * gr + k.rfar + 8.0f);   // 6 u (...) with fovT0 = 16 u
    }
    bool filled = false;
    int64_t walk = ibeg;
    for (;;) {
        int64_t base;
        if (BINR && uselist) {
            if (li < li_end) base = d.src_beg + (int64_t)d.clist[(int64_t)blockIdx.x * CLIST_MAX + li++] * TL;
            else if (tt < ntl) base = d.src_beg + tt * TL, tt += gridDim.y;
            else break;
        } else {
            base = walk;
            if (base >= iend) break;
            walk += TL;
        }
        const int cnt = (int)((iend - base) < TL ? (iend - base) : TL);  // multiple of 64
        const int nb = cnt >> 6;
        // the LDS and global addresses of the fill and the classification are formed HERE, once per tile: hoisted out of the
        // tile loop (they only depend on the thread's number) they cost six registers, which were spilled to scratch memory
        // (the thread's number put together again from the wave's and the lane's: kept from the start of the kernel it was the one
        // register spilled to scratch memory, and reloaded - with a wait - in front of every tile)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        int tid = (wave << 6) | ln;
        asm volatile("" : "+v"(tid));
        if (BINR) {
            // is any batch of this tile within the far-field radius of any receiver of the group?  (every wave
            // evaluates the same 16 circles, so the answer - and the barriers below - are uniform in the workgroup)
            const float4 bb = d.bnd[(base >> 6) + ((lane & (NBT - 1)) < nb ? (lane & (NBT - 1)) : 0)];
            float gxv = gx, gyv = gy, grv = gr;   // (copied from the scalar registers here, per tile)
            asm volatile("" : "+v"(gxv), "+v"(gyv), "+v"(grv));
            const float ex = (bb.x - og.x) - gxv, ey = (bb.y - og.y) - gyv;
            const float reach = k.rfar + bb.z + grv;
            if (ballot1(ex * ex + ey * ey <= reach * reach) == 0ull) continue;
            if (filled) __syncthreads();
            fill_tile(base, cnt, tid);
            __syncthreads();
            filled = true;
        } else if (base != ibeg) {
            __syncthreads();
            fill_tile(base, cnt, tid);
            __syncthreads();
        }
        // classification of the tile's (at most 16) batches for all four receivers in one pass:
        // lanes 16u .. 16u+15 hold receiver u, lane & 15 selects the batch
        static_assert((NBT == 16 || NBT == 32) && RPW == 4, "one classification pass covers 4 receivers x 16 batches, or 2 x 32");
        unsigned long long cand_all = ~0ull, inside_all = 0ull;
        if (CLASSIFY) {
            constexpr int PASSES = RPB >= CW * RPP ? RPB / (CW * RPP) : 1;   // four receivers x 16 batches per pass (two x 32)
#pragma unroll
            for (int ps = 0; ps < PASSES; ps++) {
                if (RPB < CW * RPP && wave * RPP >= RPB) break;   // (8 receivers: two waves classify)
                const int r0 = (wave * PASSES + ps) * RPP;   // first receiver of this pass
                bool out, in;
                classify_batch<P2R>(k, rrec[r0 + ln / NBT], tbnd[ln & (NBT - 1)], out, in);
                const bool valid = (ln & (NBT - 1)) < nb;
                cand_all = ballot1(valid & !out);
                inside_all = ballot1(valid & in);
                if (DYN && ln < RPP) {
                    if (NBT > 16) {
                        bmask[r0 + ln] = (unsigned)(cand_all >> ((NBT & 63) * ln)) & NBMASK;
                        bmask_in[r0 + ln] = (unsigned)(inside_all >> ((NBT & 63) * ln)) & NBMASK;
                    } else {
                        bmask[r0 + ln] = ((unsigned)(cand_all >> (16 * ln)) & 0xFFFFu) |
                                         (((unsigned)(inside_all >> (16 * ln)) & 0xFFFFu) << 16);
                    }
                }
            }
            if (DYN) __syncthreads();
        }
        // The near pairs this wave has noted in this tile (receiver, tile index), one per lane: the pair from the precise
        // records (precise_delta: two levels of global loads - once per wave and tile, not per receiver), field of view
        // decided on the precise pair, MINUS the pair as the fast path evaluated it (scene coordinates from the tile);
        // the differences are added to the receivers' sums in list order (fixed: bit-reproducible).
        auto near_drain = [&]() {
            if (nlen == 0) return;
            const Dev &dc = cold_args();
            const PairConsts &kc = dc.pc;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const unsigned ent = nlist[wave][lane & (NCAP - 1)];
            bool act = lane < nlen;
            // a pair can be noted twice - within rounding of a field-of-view edge (the test) AND near (the field): once is enough
            for (int i = 0; i + 1 < nlen; i++) {
                const unsigned ei = (unsigned)__builtin_amdgcn_readlane((int)ent, i);
                act = act & !((lane > i) & (ent == ei));
            }
            const int urn = lane < nlen ? (int)(ent >> 16) : 0;      // (a duplicate keeps its receiver: this wave's own, it adds 0 to it)
            const int idx = lane < nlen ? (int)(ent & 0xFFFFu) : 0;
            const int32_t ar = ragent[urn];
            act = act & (ar >= 0);                                   // (a clamped duplicate's results are not stored)
            const int32_t arc = act ? ar : 0;
            const int32_t as = act ? dc.perm[base + idx] : arc;
            // intersection.py:690-745 on the precise pair (positions to ~2e-7 m, headings to 6e-8 rad).  A pair that is within
            // rounding of an edge even so (one in ~1e7) is handed to the per-agent kernel with the source's fp64 position, to
            // be decided as the reference decides it - fp64 atan2 -> limitAngle -> angleDifference - and put right there
            // (csf_dev.h: EdgeRec; csf_agent.hip: COMBINE).  A rank of a sharded run holds only the 16-byte record of a foreign
            // source: there the precise pair decides.
            float dx, dy, F, hx, hy;
            float4 qs;
            precise_delta(dc, arc, as, dx, dy, qs);
            const float4 qr = rrec[urn];
            const Recv rr{qr.x, qr.y, qr.z, qr.w};
            const float r2p = dx * dx + dy * dy;
            bool edge;
            bool seen = tracked_precise<P2R>(kc, k.chs, rr, dx, dy, r2p, edge) & act & (as != arc);
            edge = edge & act & (as != arc);
            // np.sign(phi) undecided even here: evaluated with +1, and with -1 for the per-agent kernel to choose from
            const bool side = side_undecided(kc, qs, dx, dy, r2p) & act & (as != arc) & (dc.edge != nullptr);   // (nobody to hand it to: the pair's own sign)
            field_twod(k, rr, qs, dx, dy, fmaxf(r2p, 1e-30f), F, hx, hy, side ? 1.0f : 0.0f);
            if (__builtin_expect((ballot1(edge) | ballot1(side)) != 0ull, 0)) {
                if ((edge | (side & seen)) && dc.edge != nullptr) {
                    float F2 = 0.0f, h2x = 0.0f, h2y = 0.0f;
                    if (side) field_twod(k, rr, qs, dx, dy, fmaxf(r2p, 1e-30f), F2, h2x, h2y, -1.0f);
                    edge_handover(dc, arc, as, SEG ? seg_hfov : dc.p.hfov, F * hx, F * hy, seen, side, F2 * h2x, F2 * h2y);
                }
            }
            F = seen ? F : 0.0f;
            float fx = F * hx, fy = F * hy;
            {   // ... minus what the fast path added for it (every noted pair was kept and evaluated there)
                const float sx = rr.x - tx[idx], sy = rr.y - ty[idx];
                field_twod<false>(k, rr, make_float4(0.f, 0.f, tc[idx], ts[idx]), sx, sy, fmaxf(sx * sx + sy * sy, 1e-30f), F, hx, hy);
                F = act ? F : 0.0f;
                fx -= F * hx;
                fy -= F * hy;
            }
            for (int i = 0; i < nlen; i++) {
                const int ui = __builtin_amdgcn_readlane(urn, i);
                const float vx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fx), i));
                const float vy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fy), i));
                if (DYN) {
                    if (lane == 0) racc[0][ui] += vx, racc[1][ui] += vy;
                } else {
#pragma unroll
                    for (int q = 0; q < RPW; q++) {
                        const bool mine = (lane == 0) & (ui == wave * RPW + q);
                        ax[q] += mine ? vx : 0.0f;
                        ay[q] += mine ? vy : 0.0f;
                    }
                }
            }
            evals += (unsigned)nlen;
            nlen = 0;
        };
        // About half of the (receiver, tile) visits at N = 16 384 are EMPTY - the tile lies behind the receiver or beyond the
        // far-field radius - and a visit costs two LDS round trips (the claim, the mask) and a column sum before any source is
        // looked at.  COMPACT (receivers in slot order): every wave reads ALL the workgroup's batch masks once per tile (lane r:
        // receiver r) and only the receivers some batch can act on are handed out: the k-th claim is the k-th such receiver.
        // With receivers in binned order (neighbours: a tile concerns all of them or, mostly, none - and then it was never
        // loaded) an empty visit is just left at once.
        constexpr bool COMPACT = DYN && CLASSIFY && !BINR;
        const unsigned live = nb >= NBT ? NBMASK : ((1u << (nb & 31)) - 1u);
        unsigned mymask = 0u;
        unsigned long long wanted = 0ull;
        if (COMPACT) {
            mymask = bmask[ln & (RPB - 1)];
            wanted = ballot1((mymask & live) != 0u) & (RPB >= 64 ? ~0ull : ((1ull << (RPB & 63)) - 1ull));
        }
        const int n_wanted = COMPACT ? __builtin_popcountll(wanted) : RPB;
        // one VISIT: receiver ur of the workgroup (batch masks bm / bm_in) against this tile
        auto visit = [&](int ur, unsigned bm, unsigned bm_in, int uu) {
            const int u = DYN ? 0 : uu;               // accumulator slot
            cur_recv = ur;
            {
                const float4 q = rrec[ur];
                // the receiver in SCALAR registers (round 3): every packed instruction of the test and of the field takes at most
                // one of its four numbers, which the constant bus allows; as vector registers (rounds 1 and 2, when the compiler
                // answered scalar operands with copies) they were four register PAIRS - with them gone the kernel fits 64 VGPRs
                ru.x = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(q.x)));
                ru.y = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(q.y)));
                ru.c = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(q.z)));
                ru.s = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(q.w)));
            }
            unsigned cand = ((DYN && CLASSIFY) ? (NBT > 16 ? bm : (bm & 0xFFFFu)) : (unsigned)(cand_all >> (16 * u))) & live;
            const unsigned inside = ((DYN && CLASSIFY) ? (NBT > 16 ? bm_in : (bm >> 16)) : (unsigned)(inside_all >> (16 * u))) & live;
            if (REACH) {
                tests += (unsigned)WAVE * (unsigned)__builtin_popcount(cand | inside);   // every candidate batch goes through the per-lane test
                // two candidate batches at a time through the packed reach test (+ the exact field-of-view test unless
                // both are wholly inside); what it keeps is appended to the queue, first batch first.
                // (A single loop over both kinds of batch pairs with the records of the next pair requested ahead of the
                // queue append was tried: the compiler spilled 30 registers into the loops and the kernel ran at 180 us.)
                // test of the two batches whose records are in (sx, sy, sc, ss), append of what it keeps
                auto sift2 = [&](int b1, int b2, bool two, auto fov, v2f sx, v2f sy, v2f sc, v2f ss) {
                    constexpr bool FOV = decltype(fov)::value;
                    bool k0, k1, mg0, mg1;
                    keep_x2<FOV, P2R>(k, ru, sx, sy, sc, ss, k0, k1, mg0, mg1);
                    // (wave masks straight from the compares: a ballot of `k1 & two` would go through a vector register)
                    const unsigned long long m0 = ballot1(k0), m1 = ballot1(k1) & (0ull - (unsigned long long)two);
                    k1 = k1 & two;
                    // sources within rounding of a field-of-view edge (rare; kept): as wave masks, so that nothing but two scalars
                    // lives across the queue append
                    const unsigned long long g0 = FOV ? (ballot1(mg0) & m0) : 0ull, g1 = FOV ? (ballot1(mg1) & m1) : 0ull;
                    const int n0 = __builtin_popcountll(m0);
                    if (k0) {                       // slot = (head + length) + kept lanes below this one: v_mbcnt adds onto its operand
                        const int at = __builtin_amdgcn_mbcnt_hi((unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m0, (unsigned)(qhead + qlen)));
                        queue[wave][at & (QCAP - 1)] = (unsigned short)(4 * ((b1 << 6) + lane));
                    }
                    if (k1) {
                        const int at = __builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, (unsigned)(qhead + qlen + n0)));
                        queue[wave][at & (QCAP - 1)] = (unsigned short)(4 * ((b2 << 6) + lane));
                    }
                    qlen = __builtin_amdgcn_readfirstlane(qlen + n0 + __builtin_popcountll(m1));
                    if (FOV && __builtin_expect((g0 | g1) != 0ull, 0)) {     // ... noted: decided exactly where the noted pairs are corrected (near_drain)
                        near_note(((g0 >> lane) & 1ull) != 0ull, (b1 << 6) + lane);
                        near_note(((g1 >> lane) & 1ull) != 0ull, (b2 << 6) + lane);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    while (qlen >= CHUNK) pop(u, std::true_type{});
                };
                auto load2 = [&](int b1, int b2, bool two, auto fov) {
                    const int i0 = (b1 << 6) + lane, i1 = (b2 << 6) + lane;
                    sift2(b1, b2, two, fov, lds_pair(tx, i0, i1), lds_pair(ty, i0, i1), lds_pair(tc, i0, i1), lds_pair(ts, i0, i1));
                };
                unsigned ins = inside;
                while (__builtin_popcount(ins) >= 2) {
                    const int b1 = __builtin_ctz(ins);
                    ins &= ins - 1u;
                    const int b2 = __builtin_ctz(ins);
                    ins &= ins - 1u;
                    load2(b1, b2, true, std::false_type{});
                }
                unsigned rest = (cand & ~inside) | ins;      // partial batches (+ an odd inside one: the exact test passes all of it)
                while (rest) {
                    const int b1 = __builtin_ctz(rest);
                    rest &= rest - 1u;
                    const bool two = rest != 0u;
                    const int b2 = two ? __builtin_ctz(rest) : b1;
                    rest &= rest - 1u;                       // (0 & anything: stays 0)
                    load2(b1, b2, two, std::true_type{});
                }
                cand = 0u;
            }
            // batches that are entirely inside the field of view need neither the test nor the queue: two at a
            // time they go straight into the packed field evaluation (one batch per half of the register pairs)
            if (!REACH) {
                unsigned ins = inside;
                while (__builtin_popcount(ins) >= 2) {
                    const int b1 = __builtin_ctz(ins);
                    ins &= ins - 1u;
                    const int b2 = __builtin_ctz(ins);
                    ins &= ins - 1u;
                    const int i0 = (b1 << 6) + lane, i1 = (b2 << 6) + lane;
                    unsigned long long n0, n1;
                    field_twod_x2<true, true>(k, ru, lds_pair(tx, i0, i1), lds_pair(ty, i0, i1), lds_pair(tc, i0, i1),
                                              lds_pair(ts, i0, i1), true, true, ax[u], ay[u], &n0, &n1);
                    if (__builtin_expect((n0 | n1) != 0ull, 0)) {
                        near_note(((n0 >> lane) & 1ull) != 0ull, i0);
                        near_note(((n1 >> lane) & 1ull) != 0ull, i1);
                    }
                    evals += 2 * WAVE;
                }
                cand = (cand & ~inside) | ins;  // an odd one out takes the queue together with the partial batches
            }
            while (cand) {
                const int b = __builtin_ctz(cand);
                cand &= cand - 1u;
                const int tb = (b << 8) + 4 * lane;   // byte offset of this lane's record in the tile arrays
                if ((inside >> b) & 1u) {
                    queue[wave][(qhead + qlen + lane) & (QCAP - 1)] = (unsigned short)tb;
                    qlen = __builtin_amdgcn_readfirstlane(qlen + WAVE);
                } else {
                    const float dx = ru.x - *(const float *)((const char *)tx + tb), dy = ru.y - *(const float *)((const char *)ty + tb);
                    bool in, lt;   // kept: tracked, or within rounding of an edge - then noted, and decided exactly in near_drain
                    tracked_c<P2R>(k, ru, dx, dy, in, lt);
                    tests += (unsigned)WAVE;
                    const unsigned long long m = ballot1(in), mg = ballot1(lt) & m;
                    if (in) {
                        const int pre = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                        queue[wave][(qhead + qlen + pre) & (QCAP - 1)] = (unsigned short)tb;
                    }
                    qlen = __builtin_amdgcn_readfirstlane(qlen + __builtin_popcountll(m));
                    if (__builtin_expect(mg != 0ull, 0)) near_note(((mg >> lane) & 1ull) != 0ull, (b << 6) + lane);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (qlen >= CHUNK) pop(u, std::true_type{});
            }
            // the queue holds indices into this tile for this receiver: drain it before either changes
            while (qlen >= CHUNK) pop(u, std::true_type{});
            if (qlen > WAVE) pop(u, std::false_type{});
            else if (qlen > 0) pop_tail(u);
            if (DYN) {  // column sum of this receiver: x in the lower half of the wave, y in the upper, then within halves
                float v = swap_add32(ax[0], ay[0]);
                v += dpp<DPP_ROW_ROR8>(v);
                v += dpp<DPP_XOR1>(v);
                v += dpp<DPP_XOR2>(v);
                v += dpp<DPP_HALF_MIRROR>(v);                              // every lane: the sum of its row of 16
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                v = __uint_as_float(r[0]) + __uint_as_float(r[1]);          // rows 0+1 (x) in lanes < 32, rows 2+3 (y) beyond
                if ((lane & 31) == 0) racc[lane >> 5][ur] += v;  // one wave per receiver and tile: fixed order, no atomics
                ax[0] = ay[0] = 0.0f;
            }
        };
        if (DYN) {
            // receivers are claimed from the workgroup's counter until none is left; the wave's list of noted pairs is
            // corrected when the tile is done - or, in a crowd, as soon as it is half full, between two receivers (every
            // noted pair then belongs to a receiver whose column sum has been added: the order of the additions is fixed)
            bool more = true;
            do {
                unsigned bm = 0u, bm_in = 0u;
                int got = 0;
                if (lane == 0) got = atomicAdd(&next_recv, 1);
                int ur = __builtin_amdgcn_readfirstlane(got);
                bool skip = false;
                if (COMPACT) {                        // the ur-th set bit of `wanted`: the lane whose bit it is has `ur` set bits below it
                    more = ur < n_wanted;
                    if (more) {
                        const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(wanted >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)wanted, 0u));
                        const unsigned long long hit = ballot1(below == ur) & wanted;   // (the lanes from the ur-th set bit up to the next one, masked)
                        ur = __builtin_ctzll(hit);
                        bm = (unsigned)__builtin_amdgcn_readlane((int)mymask, ur);
                        if (NBT > 16) bm_in = (unsigned)__builtin_amdgcn_readfirstlane((int)bmask_in[ur]);   // (with the receiver's record: one round trip)
                    }
                } else {
                    more = ur < RPB;
                    if (more && CLASSIFY) {
                        bm = (unsigned)__builtin_amdgcn_readfirstlane((int)bmask[ur]);
                        skip = (bm & live) == 0u;     // no batch of this tile can act on this receiver: nothing to add
                        if (!skip && NBT > 16) bm_in = (unsigned)__builtin_amdgcn_readfirstlane((int)bmask_in[ur]);
                    }
                }
                if (more && !skip) visit(ur, bm, bm_in, 0);
                if (!more || nlen >= NCAP / 2) near_drain();
            } while (more);
        } else {
#pragma unroll
            for (int uu = 0; uu < RPW; uu++) visit(wave * RPW + uu, 0u, 0u, uu);
            near_drain();
        }
    }
    if (DYN) {
        const float4 *const chase = cold_args().part4;   // (uniform; NULL but for the launches the per-agent kernel runs beside)
        if (chase != nullptr) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's hand-over stores have landed (edge_handover)
        __syncthreads();
        const int te = (wave << 6) | lane;     // (threadIdx.x, not kept alive across the kernel)
        if (te < RPB) {
            const int64_t a = ragent[te];
            if (a >= 0) {
                if (chase != nullptr) {
                    // a per-agent wave of another XCD reads it within this launch: ONE 16-byte write-through store, the tick's tag
                    // beside the sums - nobody waits for it here and nobody signals (a wave that waited for its stores before an
                    // arrival counter kept its workgroup's slot ~1.5 us longer: + 2 us on the launch)
                    const Dev &dc = cold_args();
                    st_granule16(&dc.part4[(int64_t)(d.part_base + by) * d.cap + a], racc[0][te], racc[1][te], __uint_as_float(dc.chase_tag), 0.0f);
                } else {
                    d.part[(int64_t)(d.part_base + by) * d.cap + a] = make_float2(racc[0][te], racc[1][te]);
                }
            }
        }
        if (chase != nullptr && wave == 0 && lane == 0) {
            const Dev &dc = cold_args();
            atomicAdd(&dc.chase_misc[0], 1u);                                      // (the gate of the per-agent launch: a count, no hand-off)
            if (dc.chase_clock != nullptr) atomicMax(dc.chase_clock + 8 * dc.chase_slot + 1, (unsigned long long)wall_clock64());
        }
    } else {
        reduce_store(d, j0, lane, ax, ay, &ragent[wave * RPW]);
    }
    if (ndrop != 0u && lane == 0) atomicAdd(cold_args().near_dropped, ndrop);
    if (d.pair_count != nullptr && lane == 0) {
        atomicAdd(d.pair_count, (unsigned long long)evals + (unsigned long long)CHUNK * (pops & 0xFFFFu));
        atomicAdd(d.pair_count + 1, (unsigned long long)tests);
        atomicAdd(d.pair_count + 2, (unsigned long long)(pops & 0xFFFFu));
        atomicAdd(d.pair_count + 3, (unsigned long long)(pops >> 16));
    }
    if (d.trace && lane == 0) {   // tools/block_trace.py: when did every wave run, and where
        uint64_t *o = d.trace + 3 * (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * CW + wave);
        o[0] = t_start;
        o[1] = wall_clock64();
        o[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11))                 // HW_REG_HW_ID
               | ((uint64_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);  // HW_REG_XCC_ID
    }
}

// ---- simple kernel: every pair evaluated, masked afterwards (csf_pair_dev.h: plain_pair_sums) --
template <int FIELD, bool P2R, bool HET = false>
__global__ __launch_bounds__(BLOCK) void pair_kernel(const Dev d) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t j0 = d.lo + ((int64_t)blockIdx.x * WPB + wave) * RPW;
    float ax[RPW], ay[RPW];
    plain_pair_sums<FIELD, P2R, HET>(d, j0, lane, ax, ay);
    reduce_store(d, j0, lane, ax, ay);
}

// ---- Bicycle field on binned records: whole batches outside the field of view are skipped --------------------
// The older elliptic field (vehicle.py:1054-1147) costs ~30 instructions per pair, so there is no sifting and no
// queue here: batches are classified exactly as in pair_cull_kernel (bounding circle against the field-of-view cone,
// priority-to-the-right side test), the outside ones (62 % at hfov = 2 pi / 3) are skipped, the others are evaluated
// lane by lane, with the exact mask of intersection.py:690-745 unless the whole batch is inside.
// The field decays like exp(-0.084 rho) at worst (p_decay = 5 m): no far-field cull.
// RW receivers per wave: 4, or 8 where the grid stays large (one tile fill for twice the receivers)
template <bool P2R, int RW>
__global__ __launch_bounds__(BLOCK) void pair_bike_kernel(const Dev d) {
    __shared__ float4 tile[TILE];
    __shared__ float2 tile2[TILE];
    __shared__ float4 tbnd[TILE / WAVE];
    __shared__ float4 rrec[WPB * RW];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t j0 = d.lo + ((int64_t)blockIdx.x * WPB + wave) * RW;
    int64_t ibeg, iend;
    source_chunk(d, ibeg, iend);
    if (d.bnd_next != nullptr && blockIdx.y == 0) {   // the next tick's bounding circles (see pair_cull_kernel)
        for (int64_t b = (d.src_beg >> 6) + (int64_t)blockIdx.x * WPB + wave; b * WAVE < d.n_src; b += (int64_t)gridDim.x * WPB)
            batch_circle(d, b, lane, d.bnd_margin, d.bnd_next);
    }
    if (ibeg >= iend) return;
    if (threadIdx.x < WPB * RW) {
        const int64_t j = d.lo + (int64_t)blockIdx.x * WPB * RW + threadIdx.x;
        const int64_t jc = j < d.hi ? j : d.hi - 1;       // clamp: results of the duplicates are not stored
        rrec[threadIdx.x] = d.recg[jc];                    // scene coordinates, identical with the slot's entry of recs
    }
    float ax[RW], ay[RW];
#pragma unroll
    for (int u = 0; u < RW; u++) ax[u] = ay[u] = 0.0f;
    PairConsts k = d.pc;
    asm volatile("" : "+v"(k.lf0), "+v"(k.kexp), "+v"(k.chs), "+v"(k.ipd));
    static_assert(TILE / WAVE <= 16 && RW % 4 == 0, "one classification pass covers 4 receivers x 16 batches");
    for (int64_t base = ibeg; base < iend; base += TILE) {
        const int cnt = (int)((iend - base) < TILE ? (iend - base) : TILE);  // multiple of 64
        const int nb = cnt >> 6;
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += BLOCK) {
            tile[t] = d.recs[base + t];
            tile2[t] = d.recs2[base + t];
        }
        if ((int)threadIdx.x < nb) tbnd[threadIdx.x] = d.bnd[(base >> 6) + threadIdx.x];
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < RW / 4; ps++) {
        unsigned long long cand_all, inside_all;
        {
            bool out, in;
            classify_batch<P2R>(k, rrec[wave * RW + 4 * ps + (lane >> 4)], tbnd[lane & 15], out, in);
            const bool valid = (lane & 15) < nb;
            cand_all = ballot1(valid & !out);
            inside_all = ballot1(valid & in);
        }
#pragma unroll
        for (int uq = 0; uq < 4; uq++) {
            const int u = 4 * ps + uq;
            Recv r;
            {
                const float4 q = rrec[wave * RW + u];
                r.x = q.x, r.y = q.y, r.c = q.z, r.s = q.w;
                asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.c), "+v"(r.s));
            }
            unsigned cand = (unsigned)(cand_all >> (16 * uq)) & 0xFFFFu;
            const unsigned inside = (unsigned)(inside_all >> (16 * uq)) & 0xFFFFu;
            while (cand) {
                const int b = __builtin_ctz(cand);
                cand &= cand - 1u;
                const int t = (b << 6) + lane;
                const float4 q = tile[t];
                const float2 qb = tile2[t];
                float dx = r.x - q.x, dy = r.y - q.y;                // vehicle.py:1615-1616
                float r2 = dx * dx + dy * dy;
                const bool whole = (inside >> b) & 1u;               // (uniform) every source of the batch is tracked
                bool mg = false, edge = false;
                int32_t esrc = 0;
                bool lt = false;
                bool in = whole ? true : tracked_m<P2R>(k, k.chs, r, dx, dy, r2, mg, lt);
                mg = mg & lt;
                {   // near sources, and sources within rounding of a field-of-view edge (both rare): (dx, dy) from the precise
                    // records (precise_delta) and the decision on them
                    const bool fix = (r2 < d.pc.rnear2) | mg;
                    const int64_t jr = j0 + u;
                    if (ballot1(fix) != 0ull && jr < d.hi) {
                        const int32_t as = fix ? d.perm[base + t] : (int32_t)jr;   // (the receiver itself: a zero, masked below)
                        float px, py;
                        float4 qs;
                        precise_delta(d, (int32_t)jr, as, px, py, qs);
                        dx = fix ? px : dx;
                        dy = fix ? py : dy;
                        r2 = dx * dx + dy * dy;
                        if (fix) in = tracked_precise<P2R>(k, k.chs, r, dx, dy, r2, edge) & (as != (int32_t)jr);
                        edge = edge & fix & (as != (int32_t)jr);
                        esrc = as;
                    }
                }
                float F, gx, gy;
                r2 = fmaxf(r2, 1e-30f);      // self / coincident pair: keep every intermediate finite (F is masked)
                field_bicycle(k, q, qb, dx, dy, r2, F, gx, gy);
                if (d.edge != nullptr && ballot1(edge) != 0ull) {   // undecidable even on the precise records: the per-agent kernel decides
                    if (edge) edge_handover(d, (int32_t)(j0 + u), esrc, d.p.hfov, F * gx, F * gy, in);
                }
                F = in ? F : 0.0f;
                ax[u] += F * gx;
                ay[u] += F * gy;
            }
        }
        }
    }
#pragma unroll
    for (int h = 0; h < RW / 4; h++) {
        const float bx4[4] = {ax[4 * h], ax[4 * h + 1], ax[4 * h + 2], ax[4 * h + 3]};
        const float by4[4] = {ay[4 * h], ay[4 * h + 1], ay[4 * h + 2], ay[4 * h + 3]};
        reduce_store(d, j0 + 4 * h, lane, bx4, by4);
    }
}

// intersection.py:226-242: F = sum_k -F0 r_k^-sigma (v_k - p)/r_k over the polyline vertices.
// Same mapping: vertices in the lanes (two per lane, packed arithmetic), receivers wave-uniform.  NP = sigma + 1 when
// every edge shares one small integer sigma (the reference's defaults: 3; the curve scenario: 2): r^-(sigma+1) is then
// a power of rsq(r^2), one transcendental per pair instead of log2 + exp2.  NP = 0: per-vertex exponent.
// A receiver exactly on a vertex (r = 0; the reference divides by zero there) gets no force from it.
template <int NP>
__global__ __launch_bounds__(BLOCK) void road_kernel(const Dev d) {
    __shared__ float vx[TILE], vy[TILE], vf[TILE], vw[NP ? 1 : TILE];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t j0 = d.lo + ((int64_t)blockIdx.x * WPB + wave) * RPW;
    // the receivers' scene coordinates as two floats each, hi + lo = origin + offset exactly (csf_dev.h: rec, rorg): the
    // vertices of a tile are offsets from the tile's origin, and (hi - origin) + lo is the receiver's offset from it
    float rh[RPW][2], rl[RPW][2];
#pragma unroll
    for (int u = 0; u < RPW; u++) {
        const int64_t j = j0 + u < d.hi ? j0 + u : d.hi - 1;  // clamp: results of the duplicates are not stored
        const float4 q = d.rec[j];
        const float2 o = d.rorg[j];
        two_sum(o.x, q.x, rh[u][0], rl[u][0]);
        two_sum(o.y, q.y, rh[u][1], rl[u][1]);
    }
    Recv r[RPW];
    v2f ax[RPW], ay[RPW];
#pragma unroll
    for (int u = 0; u < RPW; u++) ax[u] = ay[u] = v2f{0.f, 0.f};
    static_assert(TILE == 1024, "the origins of the road vertices are per tile of 1024");
    for (int64_t base = 0; base < d.nv_pad; base += TILE) {
        const int cnt = (int)((d.nv_pad - base) < TILE ? (d.nv_pad - base) : TILE);  // multiple of 64
        const float2 ot = d.rvo[base >> 10];
#pragma unroll
        for (int u = 0; u < RPW; u++) {
            r[u].x = (rh[u][0] - ot.x) + rl[u][0];
            r[u].y = (rh[u][1] - ot.y) + rl[u][1];
            asm volatile("" : "+v"(r[u].x), "+v"(r[u].y));    // stay in VGPRs
        }
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += BLOCK) {
            const float4 v = d.rv[base + t];  // (x, y, -F0, -(sigma+1)/2); padding has F0 = 0
            vx[t] = v.x;
            vy[t] = v.y;
            vf[t] = v.z;
            if (!NP) vw[t] = v.w;
        }
        __syncthreads();
        for (int t = lane; t < cnt; t += 2 * WAVE) {
            const bool two = t + WAVE < cnt;
            const int t2 = two ? t + WAVE : t;
            const v2f px = lds_pair(vx, t, t2), py = lds_pair(vy, t, t2);
            v2f pf = lds_pair(vf, t, t2);
            pf.y = two ? pf.y : 0.0f;
            v2f pw{0.f, 0.f};
            if (!NP) pw = lds_pair(vw, t, t2);
#pragma unroll
            for (int u = 0; u < RPW; u++) {
                const v2f ex = px - r[u].x, ey = py - r[u].y;       // :235-236 (numerators)
                const v2f r2 = ex * ex + ey * ey;                   // :231-234
                v2f m;
                if (NP) {
                    v2f inv = rsq2(r2);
                    inv = __builtin_elementwise_min(inv, v2f{1e6f, 1e6f});   // r = 0: finite, times ex = ey = 0
                    const v2f i2 = inv * inv;
                    m = NP == 2 ? i2 : NP == 3 ? i2 * inv : NP == 4 ? i2 * i2 : NP == 5 ? i2 * i2 * inv : i2 * i2 * i2;
                } else {
                    v2f lg = pw * v2f{fast_log2(r2.x), fast_log2(r2.y)};
                    lg = __builtin_elementwise_min(lg, v2f{120.f, 120.f});   // r = 0: finite, times ex = ey = 0
                    m = v2f{fast_exp2(lg.x), fast_exp2(lg.y)};
                }
                m = m * pf;                                         // -F0 r^-(sigma+1)    :238
                ax[u] = m * ex + ax[u];                             // :239-240
                ay[u] = m * ey + ay[u];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < RPW; u++) {
        float sx = ax[u].x + ax[u].y, sy = ay[u].x + ay[u].y;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sx += __shfl_xor(sx, o, WAVE);
            sy += __shfl_xor(sy, o, WAVE);
        }
        if (lane == 0 && j0 + u < d.hi) d.froad[j0 + u] = make_float2(sx, sy);
    }
}

// Known-answer entry: m independent (source, receiver) pairs through the same device functions.
// Even slots go through the scalar field, odd slots through lane .y of the packed one (both are product code).
__global__ void pair_kat_kernel(const Dev d, const float4 *src, const float2 *src2, const float4 *recv,
                                int64_t m, int apply_fov, float2 *out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const PairConsts k = d.pc;
    float4 q = src[t], rr = recv[t];
    Recv r{rr.x, rr.y, rr.z, rr.w};
    float dx = r.x - q.x, dy = r.y - q.y, r2 = dx * dx + dy * dy;
    bool in = r2 > 0.f;
    if (apply_fov) in = k.p2r ? tracked<true>(k.chs, r, dx, dy, r2) : tracked<false>(k.chs, r, dx, dy, r2);
    r2 = fmaxf(r2, 1e-30f);
    float fx = 0.f, fy = 0.f;
    if (d.p.model == CSF_BICYCLE) {
        float F, gx, gy;
        field_bicycle(k, q, src2[t], dx, dy, r2, F, gx, gy);
        fx = F * gx;
        fy = F * gy;
    } else if (t & 1) {
        // the packed field as the cull-first kernel uses it: a pair it flags (near, or np.sign(phi) within rounding of phi = 0)
        // is evaluated once more by the unpacked field - there from the precise records - and that result counts
        float px = 0.f, py = 0.f;
        unsigned long long n0 = 0ull, n1 = 0ull;
        if (t & 2) field_twod_x2<true, true>(k, r, v2f{q.x, q.x}, v2f{q.y, q.y}, v2f{q.z, q.z}, v2f{q.w, q.w}, true, true, px, py, &n0, &n1);
        else field_twod_x2<false, true>(k, r, v2f{r.x, q.x}, v2f{r.y, q.y}, v2f{1.f, q.z}, v2f{0.f, q.w}, false, true, px, py, &n0, &n1);
        fx = (t & 2) ? 0.5f * px : px;
        fy = (t & 2) ? 0.5f * py : py;
        if ((n1 >> (threadIdx.x & 63)) & 1ull) {
            float F, gx, gy;
            field_twod(k, r, q, dx, dy, r2, F, gx, gy);
            fx = F * gx;
            fy = F * gy;
        }
    } else {
        float F, gx, gy;
        field_twod(k, r, q, dx, dy, r2, F, gx, gy);
        fx = F * gx;
        fy = F * gy;
    }
    if (!in) fx = fy = 0.f;
    out[t] = make_float2(fx, fy);
}

// get_untracked_foes (intersection.py:690-745) as the matrix the reference returns: out[i * n + j] = 1 when receiver j
// ignores source i (row = source, column = receiver; the diagonal is always 1), from the same test the pair kernels apply
__global__ void untracked_kernel(const Dev d, uint8_t *out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = d.n_live;
    if (t >= n * n) return;
    const int64_t i = t / n, j = t - i * n;
    const int64_t ai = d.order ? d.order[i] : i, aj = d.order ? d.order[j] : j;
    const csf_params &ps = d.ptab[d.cls[ai]];                   // the hfov of the source's parameter set (:733-735)
    bool in;
    if (d.state_current) {   // as the reference decides it: fp64, atan2 -> limitAngle -> angleDifference on the fp64 state
        in = (i != j) && !untracked_exact(d, ai, aj, ps.hfov, d.pc.p2r != 0);
    } else {                 // a rank of a sharded run: the precise records of the pair
        const float4 q = d.rec[ai], rr = d.rec[aj];
        const float2 qo = d.rorg[ai], ro = d.rorg[aj];         // the records are offsets from origins of their own
        const Recv r{rr.x, rr.y, rr.z, rr.w};
        const float dx = (r.x - q.x) + (ro.x - qo.x), dy = (r.y - q.y) + (ro.y - qo.y), r2 = dx * dx + dy * dy;
        const float chs = d.pctab[d.cls[ai]].chs;
        in = (i != j) & (d.pc.p2r ? tracked<true>(chs, r, dx, dy, r2) : tracked<false>(chs, r, dx, dy, r2));
    }
    out[t] = in ? 0 : 1;
}

void launch_untracked(const Dev &d, uint8_t *out, hipStream_t st) {
    const int64_t m = d.n_live * d.n_live;
    if (m <= 0) return;
    hipLaunchKernelGGL(untracked_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, d, out);
}

static dim3 recv_grid(const Dev &d, int split, int per_block_recv = WPB * RPW) {
    int64_t nloc = d.hi - d.lo;
    int64_t per_block = per_block_recv;
    return dim3((unsigned)((nloc + per_block - 1) / per_block), (unsigned)split, 1);
}

// every launch of this file: optional events take the kernel's own start / end time stamps (csf_dev.h)
#define CSF_LAUNCH(kernel, grid) hipExtLaunchKernelGGL(kernel, grid, dim3(BLOCK), 0, st, t0, t1, 0, d)

template <bool P2R, bool CLASSIFY, bool BINR>
static void launch_cull_dyn(const Dev &d, hipStream_t st, hipEvent_t t0, hipEvent_t t1) {
    if (CLASSIFY && d.dyn_recv && d.pc.reach) {      // the default: per-pair reach test in front of the field
        if (d.rpb == 16 && d.wide && CLASSIFY) {
            hipExtLaunchKernelGGL((pair_cull_kernel<P2R, CLASSIFY, BINR, true, 16, CLASSIFY, CLASSIFY ? 2 * WPB : WPB>), recv_grid(d, d.n_split, 16),
                                  dim3(2 * BLOCK), 0, st, t0, t1, 0, d);
        } else if (d.rpb == 32 && d.wide && CLASSIFY) {   // 8 waves, tiles of 2048 sources (csf_engine.hip set_chunks: chunks of 32 batches)
            hipExtLaunchKernelGGL((pair_cull_kernel<P2R, CLASSIFY, BINR, true, 32, CLASSIFY, CLASSIFY ? 2 * WPB : WPB>), recv_grid(d, d.n_split, 32),
                                  dim3(2 * BLOCK), 0, st, t0, t1, 0, d);
        } else if (d.rpb == 32) CSF_LAUNCH((pair_cull_kernel<P2R, CLASSIFY, BINR, true, 32, CLASSIFY>), recv_grid(d, d.n_split, 32));
        else if (d.rpb == 8) CSF_LAUNCH((pair_cull_kernel<P2R, CLASSIFY, BINR, true, 8, CLASSIFY>), recv_grid(d, d.n_split, 8));
        else CSF_LAUNCH((pair_cull_kernel<P2R, CLASSIFY, BINR, true, WPB * RPW, CLASSIFY>), recv_grid(d, d.n_split));
        return;
    }
    if (CLASSIFY && d.dyn_recv && d.rpb == 32 && d.wide) {   // every pair evaluated (CSF_FAR_EPS=0): the wide workgroup as well
        hipExtLaunchKernelGGL((pair_cull_kernel<P2R, CLASSIFY, BINR, true, 32, false, CLASSIFY ? 2 * WPB : WPB>), recv_grid(d, d.n_split, 32),
                              dim3(2 * BLOCK), 0, st, t0, t1, 0, d);
        return;
    }
    if (d.dyn_recv && d.rpb == 32) {
        CSF_LAUNCH((pair_cull_kernel<P2R, CLASSIFY, BINR, true, 32>), recv_grid(d, d.n_split, 32));
        return;
    }
    if (d.dyn_recv) CSF_LAUNCH((pair_cull_kernel<P2R, CLASSIFY, BINR, true>), recv_grid(d, d.n_split));
    else CSF_LAUNCH((pair_cull_kernel<P2R, CLASSIFY, BINR, false>), recv_grid(d, d.n_split));
}

// the runs of all TwoD-field parameter sets in one grid (Dev::segtab; receivers per workgroup and tile as launch_cull_dyn chooses)
void launch_pair_segments(const Dev &d, int total_by, hipStream_t st, hipEvent_t t0, hipEvent_t t1) {
    if (d.hi <= d.lo || total_by <= 0) return;
    const bool p2r = d.pc.p2r != 0;
    const int rpb = d.rpb == 32 ? 32 : 16;
    const dim3 g = recv_grid(d, total_by, rpb);
#define CSF_SEG_LAUNCH(P2R_, RPB_) hipExtLaunchKernelGGL((pair_cull_kernel<P2R_, true, false, true, RPB_, true, 2 * WPB, true>), g, dim3(2 * BLOCK), 0, st, t0, t1, 0, d)
    if (rpb == 32) {
        if (p2r) CSF_SEG_LAUNCH(true, 32);
        else CSF_SEG_LAUNCH(false, 32);
    } else {
        if (p2r) CSF_SEG_LAUNCH(true, 16);
        else CSF_SEG_LAUNCH(false, 16);
    }
#undef CSF_SEG_LAUNCH
}

static void launch_cull(const Dev &d, hipStream_t st, hipEvent_t t0, hipEvent_t t1) {
    const bool p2r = d.pc.p2r != 0;
    if (d.classify && d.recv_binned) {
        if (p2r) launch_cull_dyn<true, true, true>(d, st, t0, t1);
        else launch_cull_dyn<false, true, true>(d, st, t0, t1);
    } else if (d.classify) {
        if (p2r) launch_cull_dyn<true, true, false>(d, st, t0, t1);
        else launch_cull_dyn<false, true, false>(d, st, t0, t1);
    } else {
        if (p2r) launch_cull_dyn<true, false, false>(d, st, t0, t1);
        else launch_cull_dyn<false, false, false>(d, st, t0, t1);
    }
}

void launch_pair(const Dev &d, hipStream_t st, hipEvent_t t0, hipEvent_t t1) {
    if (d.hi <= d.lo) return;
    const dim3 g = recv_grid(d, d.n_split);
    const bool p2r = d.pc.p2r != 0;
    if (d.n_classes > 1) {                         // several parameter sets: the source's own row for every pair
        if (d.n_classes > MAX_CLASSES) return;     // (csf_set_param_classes refuses more)
        if (d.has_bike && (d.model_mask & ~1)) {   // Bicycle and other classes together: the field of the source's class
            if (p2r) CSF_LAUNCH((pair_kernel<2, true, true>), g);
            else CSF_LAUNCH((pair_kernel<2, false, true>), g);
        } else if (d.has_bike) {
            if (p2r) CSF_LAUNCH((pair_kernel<1, true, true>), g);
            else CSF_LAUNCH((pair_kernel<1, false, true>), g);
        } else {
            if (p2r) CSF_LAUNCH((pair_kernel<0, true, true>), g);
            else CSF_LAUNCH((pair_kernel<0, false, true>), g);
        }
    } else if (d.p.model == CSF_BICYCLE && d.classify && d.recs_valid) {
        if (d.rpb == 32) {
            const dim3 g8 = recv_grid(d, d.n_split, 32);
            if (p2r) CSF_LAUNCH((pair_bike_kernel<true, 8>), g8);
            else CSF_LAUNCH((pair_bike_kernel<false, 8>), g8);
        } else {
            if (p2r) CSF_LAUNCH((pair_bike_kernel<true, 4>), g);
            else CSF_LAUNCH((pair_bike_kernel<false, 4>), g);
        }
    } else if (d.p.model == CSF_BICYCLE) {
        if (p2r) CSF_LAUNCH((pair_kernel<1, true>), g);
        else CSF_LAUNCH((pair_kernel<1, false>), g);
    } else if (d.pair_variant == 1) {
        if (p2r) CSF_LAUNCH((pair_kernel<0, true>), g);
        else CSF_LAUNCH((pair_kernel<0, false>), g);
    } else {
        launch_cull(d, st, t0, t1);
    }
}

// which pair kernel launch_pair() takes for this engine (profiles and the bench line name it)
const char *pair_kernel_name(const Dev &d) {
    if (d.n_classes > 1) return "pair_kernel";
    if (d.p.model == CSF_BICYCLE) return (d.classify && d.recs_valid) ? "pair_bike_kernel" : "pair_kernel";
    if (d.pair_variant == 1) return "pair_kernel";
    return "pair_cull_kernel";
}

void launch_road(const Dev &d, hipStream_t st, hipEvent_t t0, hipEvent_t t1) {
    if (d.hi <= d.lo || d.nv == 0) return;
    if (d.rg_nx > 0) {     // a large, static network: near field summed, far field interpolated (csf_road.hip)
        Dev dd = d;        // receivers in binned order where this device bins and integrates every slot
        dd.rg_by_place = (d.recs_valid && !d.seg_keys && d.rlist == nullptr && d.lo == 0 && d.hi == d.n) ? 1 : 0;
        // (a rank's block: its receivers sorted by place - round 6: in slot order a rank's road term took 2.8 x its share, the
        // neighbours of a wave no longer sharing their cells' vertices in cache)
        if (!dd.rg_by_place && d.recs_valid && !d.seg_keys && d.rlist != nullptr && d.recv_binned) dd.rg_by_place = 2;
        launch_road_grid(dd, st, t0, t1);
        return;
    }
    const dim3 g = recv_grid(d, 1);
    switch (d.road_np) {
    case 2: CSF_LAUNCH(road_kernel<2>, g); break;
    case 3: CSF_LAUNCH(road_kernel<3>, g); break;
    case 4: CSF_LAUNCH(road_kernel<4>, g); break;
    case 5: CSF_LAUNCH(road_kernel<5>, g); break;
    case 6: CSF_LAUNCH(road_kernel<6>, g); break;
    default: CSF_LAUNCH(road_kernel<0>, g); break;
    }
}

void launch_pair_kat(const Dev &d, const float4 *src, const float2 *src2, const float4 *recv, int64_t m,
                     int apply_fov, float2 *out, hipStream_t st) {
    if (m <= 0) return;
    hipLaunchKernelGGL(pair_kat_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, d, src, src2, recv,
                       m, apply_fov, out);
}

}  // namespace csf
