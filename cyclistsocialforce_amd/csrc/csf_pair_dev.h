// csf_pair_dev.h - device helpers of the pair kernels that more than one translation unit uses (csf_pair.hip, csf_mid.hip):
// the wave's column sum, the source chunks of the grid, the precise difference of two records, and the plain all-pairs sums.
#pragma once
#include <type_traits>

#include "csf_field.h"

namespace csf {

constexpr int RPW = 4;               // receivers per wave
constexpr int WPB = 4;               // waves per workgroup
constexpr int BLOCK = WPB * WAVE;
constexpr int TILE = 1024;           // simple kernels: source records per LDS tile (16 KiB)
#ifndef CSF_TILE2
#define CSF_TILE2 1024
#endif
#ifndef CSF_CULL_WAVES
#define CSF_CULL_WAVES 8
#endif
constexpr int TILE2 = CSF_TILE2;     // culling kernel: LDS tile (records); 1024 x 8 workgroups per CU = 8 waves/SIMD (64 VGPRs, 20 228 B of LDS)
constexpr int QCAP = 256;            // queue slots per receiver (uint16 tile indices); power of two
constexpr int CHUNK = 128;           // pairs evaluated per pop: two per lane

__device__ __forceinline__ void load_receivers(const Dev &d, int64_t j0, Recv (&r)[RPW]) {
#pragma unroll
    for (int u = 0; u < RPW; u++) {
        int64_t j = j0 + u < d.hi ? j0 + u : d.hi - 1;  // clamp: results of the duplicates are not stored
        float4 q = d.rec[j];
        const float2 o = d.rorg[j];                     // (0 for unbinned populations: the plain kernel's only case)
        r[u].x = q.x + o.x;
        r[u].y = q.y + o.y;
        r[u].c = q.z;
        r[u].s = q.w;
        asm volatile("" : "+v"(r[u].x), "+v"(r[u].y), "+v"(r[u].c), "+v"(r[u].s));  // stay in VGPRs
    }
}

// column sum (intersection.py:841-843) of the 2 x RPW per-lane accumulators of a wave.  The eight values are
// reduced together: each of the first three butterfly steps hands half of the values to the partner lanes, the last
// three finish the one value a lane is left with.  gfx950 lane exchanges, no LDS and no address arithmetic
// (__shfl_xor costs four VALU instructions and a ds_bpermute each): v_permlane32_swap / v_permlane16_swap exchange the
// upper half (odd rows) of one register with the lower half (even rows) of another, so one swap and one add
// reduce two values across the halves; DPP modifiers on the adds do the rest.  Fixed order: bit-reproducible.
__device__ __forceinline__ float swap_add32(float a, float b) {   // lanes < 32: a over both halves; lanes >= 32: b
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add16(float a, float b) {   // even rows of 16: a over the row pair; odd rows: b
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float dpp(float x) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), CTRL, 0xf, 0xf, false));
}
constexpr int DPP_ROW_ROR8 = 0x128, DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141;

// reduce8: the butterfly itself - every lane of a group of 8 ends with value `idx` = 4 bit5 + 2 bit4 + bit3 of its lane number
// (2 u + component: receiver u of the wave, x or y) summed over the wave
__device__ __forceinline__ float reduce8(int lane, const float (&ax)[RPW], const float (&ay)[RPW], int &idx) {
    static_assert(RPW == 4, "the reduction below is written for eight values");
    const float v[8] = {ax[0], ay[0], ax[1], ay[1], ax[2], ay[2], ax[3], ay[3]};
    float w[4], y[2];
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = swap_add32(v[i], v[i + 4]);
#pragma unroll
    for (int i = 0; i < 2; i++) y[i] = swap_add16(w[i], w[i + 2]);
    const bool h3 = lane & 8;
    float z = (h3 ? y[1] : y[0]) + dpp<DPP_ROW_ROR8>(h3 ? y[0] : y[1]);
    z += dpp<DPP_XOR1>(z);
    z += dpp<DPP_XOR2>(z);
    z += dpp<DPP_HALF_MIRROR>(z);
    idx = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
    return z;
}

__device__ __forceinline__ void reduce_store(const Dev &d, int64_t j0, int lane, const float (&ax)[RPW],
                                             const float (&ay)[RPW], const int *agent_of = nullptr) {
    int idx;
    const float z = reduce8(lane, ax, ay, idx);
    const int u = idx >> 1;
    // agent_of: slot of each of the wave's receivers, -1 for none (LDS); else the receivers are the slots j0 + u
    const int64_t a = agent_of ? (int64_t)agent_of[u] : (j0 + u < d.hi ? j0 + u : -1);
    if ((lane & 7) == 0 && a >= 0) {
        float *dst = (float *)&d.part[(int64_t)(d.part_base + blockIdx.y) * d.cap + a];
        dst[idx & 1] = z;
    }
}

__device__ __forceinline__ void source_chunk(const Dev &d, int64_t &ibeg, int64_t &iend, int by = (int)blockIdx.y) {
    const int64_t per = d.chunk_units;  // `by` (blockIdx.y) selects a chunk of sources, in units of 64 records
    ibeg = d.src_beg + (int64_t)by * per * WAVE;
    iend = ibeg + per * WAVE;
    if (iend > d.n_src) iend = d.n_src;   // (the places behind hold sentinels only, csf_engine.hip: rebin)
}

// The kernel's argument block again, through a pointer the compiler cannot see through: what only the rare paths read
// (the precise records, the permutation, the hand-over ring ...) is then loaded where it is used.  Read through `d`, those
// loads are hoisted to the top of the kernel as loop invariants and, for want of scalar registers, parked in the lanes of a
// vector register - some fifty v_writelane per wave before the first source is looked at (65 536 waves per launch).
__device__ __forceinline__ const Dev &cold_args() {
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const Dev *)p;
}

// the wave mask of a predicate, straight from the compare that made it (HIP's __ballot takes an int: the bool is widened
// to 0 / 1 in a vector register and compared with 0 again - two vector instructions per ballot in the test loops)
__device__ __forceinline__ unsigned long long ballot1(bool b) { return __builtin_amdgcn_ballot_w64(b); }

// ---- near pairs ---------------------------------------------------------------------------------------------------
// The kernels work on fp32 positions relative to the scene origin (what the tile holds: 2^-24 of the scene extent, 8e-6 m
// at 130 m).  The field's direction turns by (position error / distance) and its decay length is as short as 0.2 m, so
// for the rare pairs closer than PairConsts::rnear (1 m) that is not enough - the reference forms x - x0 in fp64
// (vehicle.py:1615-1617).  Such a pair is CORRECTED from the precise records: every record in HBM is an offset (a few
// metres at most) from an origin of its own (a multiple of 1/4 m; csf_dev.h: rec, rorg), so
// dx = (off_r - off_s) + (org_r - org_s)  carries ~2e-7 m whatever the extent of the scene.  pair_cull_kernel: the field
// evaluation notes the near pairs it meets (two compares and a rarely taken branch per 128 pairs), and once per wave and
// tile the noted pairs are evaluated twice, one per lane - from the precise records (field of view included) and as the
// fast path saw them - and the difference is added; a list that is full leaves the rest uncorrected.  The other
// kernels evaluate every pair in one place and swap the precise (dx, dy) in.
__device__ __forceinline__ void precise_delta(const Dev &d, int32_t a_recv, int32_t a_src, float &dx, float &dy, float4 &qs) {
    const float4 qr = d.rec[a_recv];
    const float2 orr = d.rorg[a_recv];
    qs = d.rec[a_src];
    const float2 os = d.rorg[a_src];
    dx = (qr.x - qs.x) + (orr.x - os.x);
    dy = (qr.y - qs.y) + (orr.y - os.y);
}

// One LDS load per value: hipcc would otherwise merge the loads of (x, y) and (c, s) of ONE record into
// ds_read2st64_b32, whose register pair then has to be taken apart with v_mov to form the packed operands
// {x[i0], x[i1]} ... of the field (7 moves per evaluation).
typedef const volatile __attribute__((address_space(3))) float *lds_vfp;   // volatile: not merged; LDS: ds_read_b32
__device__ __forceinline__ v2f lds_pair(const float *a, int i0, int i1) {
    return v2f{*(lds_vfp)(a + i0), *(lds_vfp)(a + i1)};
}
// the same with BYTE offsets (what the queue stores: no shift between the queue read and the tile read)
__device__ __forceinline__ v2f lds_pair_b(const float *a, int o0, int o1) {
    return v2f{*(lds_vfp)((const char *)a + o0), *(lds_vfp)((const char *)a + o1)};
}

// Classification of one batch of 64 binned source records (bounding circle bb = centre, radius) against one receiver rl
// = (x, y, cos psi, sin psi).  out: no source of the batch can be tracked (or all are beyond the far-field radius);
// in: every source is tracked and none is far.  Both keep a 1e-4 margin and need the receiver outside the circle, so
// whatever they leave undecided goes to the exact per-lane test (intersection.py:690-745).
//   outside: the heading misses the circle (|beta| > alpha) and cos(|beta| - alpha) < cos(hfov/2)
//   inside:  cos(|beta| + alpha) > cos(hfov/2) with |beta| + alpha < pi (beyond, the circle reaches across the rear
//            axis and the cosine is no longer monotone: hfov > pi); any circle when hfov >= 2 pi
//   priority to the right (intersection.py:739-741): a circle wholly to the left of the heading line is outside,
//            and only one wholly to the right can be inside
//   far:     every source is beyond the radius at which the field has decayed below the resolution of the fp32
//            column sum (csf_engine.hip: far_radius; +inf for the Bicycle field)
template <bool P2R>
__device__ __forceinline__ void classify_batch(const PairConsts &k, const float4 rl, const float4 bb, bool &out, bool &in) {
    const float rc = rl.z, rs = rl.w;
    const float ex = bb.x - rl.x, ey = bb.y - rl.y;           // receiver -> centre of the batch
    const float D2 = ex * ex + ey * ey;
    const float invD = fast_rsq(fmaxf(D2, 1e-30f));
    const float sa = fminf(bb.z * invD, 1.0f);                // sin of the circle's half angle
    const float ca = fast_sqrt(fmaxf(1.0f - sa * sa, 0.0f));
    const float cb = (ex * rc + ey * rs) * invD;              // cos / |sin| of the centre's bearing
    const float off = rc * ey - rs * ex;                      // centre's offset from the heading line, left > 0
    const float sb = fabsf(off) * invD;
    // receiver outside the circle - by more than fp32 positions blur a bearing: both are off by 2^-24 of their coordinates (in
    // the frame the tile is held in), which must stay below half the 1e-4 margin of the tests below as an angle
    const float keepoff = bb.z + 0.05f + k.clsk * (fabsf(bb.x) + fabsf(bb.y) + bb.z + fabsf(rl.x) + fabsf(rl.y));
    const bool apart = D2 > keepoff * keepoff;
    const bool fov = apart & (k.fov_classify != 0);
    const float reach = k.rfar + bb.z;
    const bool far = D2 > reach * reach;
    // (k.chm, k.chp = k.ch -+ 1e-4 from the host: computed here they were loop invariants in vector registers, spilled to scratch)
    out = far | (fov & (cb < ca) & ((cb * ca + sb * sa) < k.chm));
    in = fov & !far & ((((cb * ca - sb * sa) > k.chp) & ((sb * ca + cb * sa) > 1e-4f)) | (k.full_circle != 0));
    if (P2R) {
        const float clear = bb.z * 1.0001f + 1e-4f;
        out = out | (off > clear);
        in = in & (off < -clear);
    }
}

// One source (record q in scene coordinates, second record qb, slot a_src, parameter set ks / hfov_src) against the wave's RPW
// receivers r (slots j0 ..): mask (intersection.py:690-745), field (vehicle.py:1560-1648 or 1054-1147), the rare pairs from the
// precise records or handed over, and the per-lane sums.  Shared by pair_kernel and csf_mid.hip.
// RAW: the caller holds the precise records themselves - offsets and own origins of the wave's receivers and of this lane's source
// (csf_mid.hip loads them straight from memory anyway) -, so the precise difference of a near or marginal pair is arithmetic on
// registers instead of four dependent loads in the middle of the evaluation (precise_delta).
struct PreciseRegs {
    float rx[RPW], ry[RPW], rox[RPW], roy[RPW];   // receivers: offset, origin
    float sx, sy, sox, soy;                        // this lane's source
};
template <int FIELD, bool P2R, bool RAW = false>
__device__ __forceinline__ void plain_pair_eval(const Dev &d, const PairConsts &ks, const double hfov_src, const Recv (&r)[RPW], const int64_t j0,
                                                const float4 q, const float2 qb, const int32_t a_src, float (&ax)[RPW], float (&ay)[RPW],
                                                const PreciseRegs *const pr = nullptr) {
#pragma unroll
    for (int u = 0; u < RPW; u++) {
        float dx = r[u].x - q.x, dy = r[u].y - q.y;  // vehicle.py:1615-1616
        float r2 = dx * dx + dy * dy;
        bool mg, lt, edge = false;
        bool in = tracked_m<P2R>(d.pc, ks.chs, r[u], dx, dy, r2, mg, lt);
        mg = mg & lt;
        const bool twod = !(FIELD == 1 || (FIELD == 2 && ks.ipd != 0.0f));   // (ipd: Bicycle sets only)
        bool side = false;
        {   // near sources, sources within rounding of a field-of-view edge and - TwoD field - receivers within rounding of
            // the line ahead of the source, where np.sign(phi) turns (all rare): (dx, dy) from the precise records
            // (precise_delta) and the decisions on them; this kernel's sources sit in slot order
            const bool sd = twod & (fabsf(dy * q.z - dx * q.w) < d.pc.fovT1 + d.pc.fovT0 * (0.0625f * r2 + 4.0f));   // (rho <= r2 / 16 + 4)
            const bool fix = (r2 < d.pc.rnear2) | mg | sd;
            const int64_t jr = j0 + u;
            if (ballot1(fix) != 0ull && jr < d.hi) {
                float px, py;
                float4 qs;
                const int32_t as = fix ? a_src : (int32_t)jr;
                if (RAW) {   // (the very expression of precise_delta)
                    px = (pr->rx[u] - pr->sx) + (pr->rox[u] - pr->sox);
                    py = (pr->ry[u] - pr->sy) + (pr->roy[u] - pr->soy);
                } else {
                    precise_delta(d, (int32_t)jr, as, px, py, qs);
                }
                dx = fix ? px : dx;
                dy = fix ? py : dy;
                r2 = dx * dx + dy * dy;
                if (fix) in = tracked_precise<P2R>(d.pc, ks.chs, r[u], dx, dy, r2, edge) & (as != (int32_t)jr);
                edge = edge & fix & (as != (int32_t)jr);
                side = fix & twod & (as != (int32_t)jr) & side_undecided(d.pc, q, dx, dy, r2) & (d.edge != nullptr);   // (nobody to hand it to: the pair's own sign)
            }
        }
        r2 = fmaxf(r2, 1e-30f);  // self / coincident pair: keep every intermediate finite (F is masked)
        float F, gx, gy;
        if (!twod) field_bicycle(ks, q, qb, dx, dy, r2, F, gx, gy);
        else field_twod(ks, r[u], q, dx, dy, r2, F, gx, gy, side ? 1.0f : 0.0f);
        if (d.edge != nullptr && (ballot1(edge) | ballot1(side)) != 0ull) {   // undecidable even on the precise records: the per-agent kernel decides
            if (edge | (side & in)) {
                float F2 = 0.0f, h2x = 0.0f, h2y = 0.0f;
                if (side) field_twod(ks, r[u], q, dx, dy, r2, F2, h2x, h2y, -1.0f);
                edge_handover(d, (int32_t)(j0 + u), a_src, hfov_src, F * gx, F * gy, in, side,
                              F2 * h2x, F2 * h2y);
            }
        }
        F = in ? F : 0.0f;
        ax[u] += F * gx;
        ay[u] += F * gy;
    }
}

// ---- simple kernel: every pair evaluated, masked afterwards (Bicycle field; also the TwoD field on request) --
// HET: the vehicles own different parameter sets (csf_set_param_classes).  The field of source i is evaluated with ITS
// f_0 / sigma / e (vehicle.py:1592-1612; p_0 / p_decay: 1095-1101) and masked with ITS hfov (intersection.py:733-735):
// the table of what derive_pair_consts makes of every set sits in LDS and each lane looks its source's row up.  No
// cull (the far-field bound and the batch classification are per parameter set): this is the O(N^2) path of small,
// mixed populations.
// FIELD: 0 the TwoD field (vehicle.py:1560-1648), 1 the Bicycle field (vehicle.py:1054-1147), 2 (HET only) a population of
// several vehicle classes (intersection.py:797-823: each vehicle's own calcRepulsiveForce): the field of the source's class.
constexpr int MAX_CLASSES = 256;
// plain_pair_sums: the per-lane sums of the wave's RPW receivers j0 .. over the workgroup's source chunk (sources in the lanes);
// the caller reduces and stores them (reduce_store).  Shared by pair_kernel (csf_pair.hip) and the fused tick of mid-size
// populations (csf_mid.hip).
template <int FIELD, bool P2R, bool HET>
__device__ __forceinline__ void plain_pair_sums(const Dev &d, const int64_t j0, const int lane, float (&ax)[RPW], float (&ay)[RPW]) {
    static_assert(FIELD != 2 || HET, "the field per source class needs the table of parameter sets");
    constexpr bool BICYCLE = FIELD != 0;              // the second record is read
    __shared__ float4 tile[TILE];
    __shared__ float2 tile2[BICYCLE ? TILE : 1];
    __shared__ PairConsts ctab[HET ? MAX_CLASSES : 1];
    __shared__ uint8_t tcls[HET ? TILE : 1];
    if (HET) {
        const int words = d.n_classes * (int)(sizeof(PairConsts) / 4);
        for (int w = threadIdx.x; w < words; w += BLOCK) ((uint32_t *)ctab)[w] = ((const uint32_t *)d.pctab)[w];
    }
    int64_t ibeg, iend;
    source_chunk(d, ibeg, iend);

    Recv r[RPW];
    load_receivers(d, j0, r);
#pragma unroll
    for (int u = 0; u < RPW; u++) ax[u] = ay[u] = 0.0f;
    PairConsts k = d.pc;
    asm volatile("" : "+v"(k.sg0), "+v"(k.sg1), "+v"(k.sg2), "+v"(k.sg3), "+v"(k.e0), "+v"(k.e1), "+v"(k.lf0),
                 "+v"(k.kexp), "+v"(k.chs));

    for (int64_t base = ibeg; base < iend; base += TILE) {
        int cnt = (int)((iend - base) < TILE ? (iend - base) : TILE);  // multiple of 64
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += BLOCK) {
            const float2 o = d.rorg[base + t];
            float4 q = d.rec[base + t];
            q.x += o.x, q.y += o.y;                               // scene coordinates
            tile[t] = q;
            if (BICYCLE) tile2[t] = d.rec2[base + t];
            if (HET) tcls[t] = base + t < d.n ? d.cls[base + t] : (uint8_t)0;   // (padding records are sentinels of any set)
        }
        __syncthreads();
        for (int t = lane; t < cnt; t += WAVE) {
            float4 q = tile[t];
            float2 qb = BICYCLE ? tile2[t] : make_float2(0.f, 0.f);
            const PairConsts &ks = HET ? ctab[tcls[t]] : k;   // the source's parameter set
            plain_pair_eval<FIELD, P2R>(d, ks, HET ? d.ptab[tcls[t]].hfov : d.p.hfov, r, j0, q, qb, (int32_t)(base + t), ax, ay);
        }
    }
}

}  // namespace csf
