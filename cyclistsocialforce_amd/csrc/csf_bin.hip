// csf_bin.hip — spatial binning of the source records for the cull-first pair kernel.
//
// The pair kernel streams the sources in batches of 64 (one per lane).  When consecutive records are spatial
// neighbours, a whole batch can be classified against a receiver's field-of-view cone from its bounding
// circle alone (csf_pair.hip): entirely outside -> skipped, entirely inside -> queued without per-lane tests.
// This file provides the order and the circles:
//   * every REBIN ticks: Hilbert index of each record's cell (0.5 m) -> stable radix sort (hipCUB) -> perm[];
//   * every record's own origin (where the road user is now, rounded to 1/4 m), which its fp32 position is an offset
//     from until the next re-binning (rebase_kernel): a position resolves to 2^-24 of a few metres, not of the scene;
//   * bounding circle of each batch of 64 records in perm order: right after a re-sort by rebase_kernel, otherwise
//     by the pair kernel itself, which emits the circles of the NEXT tick from this tick's records grown by the
//     largest possible displacement of one tick (positions move every tick, the order only drifts, so
//     correctness never depends on how fresh perm is).
// Nothing here changes results: the per-pair test of intersection.py:690-745 stays exact; the sort is stable
// and deterministic, so runs are bit-reproducible.
#include <hipcub/hipcub.hpp>

#include "csf_dev.h"

namespace csf {

// Hilbert index of a 16-bit cell coordinate pair: consecutive indices are always adjacent cells, so batches of
// 64 consecutive records are compact (mean bounding radius 11.7 m at N = 16 384 in 200 m x 200 m against 15.9 m
// for Morton order, which jumps; fewer batches straddle a field-of-view edge).
__device__ __forceinline__ uint32_t hilbert16(uint32_t x, uint32_t y) {
    uint32_t d = 0;
    for (uint32_t s = 1u << 15; s > 0; s >>= 1) {
        const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
        d += s * s * ((3u * rx) ^ ry);
        if (ry == 0) {
            if (rx == 1) {
                x = s - 1 - x;
                y = s - 1 - y;
            }
            const uint32_t t = x;
            x = y;
            y = t;
        }
    }
    return d;
}

__global__ void keys_kernel(const Dev d, uint32_t *keys, int32_t *vals) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= d.n_pad) return;
    const float4 q = d.rec[a];
    uint32_t key = 0xFFFFFFFFu;  // sentinels (padding, shard holes) sort to the end
    if (fabsf(q.x) < 1e9f && fabsf(q.y) < 1e9f) {
        const float2 o = d.rorg[a];  // the record is an offset from the origin of the batch it was binned into last time
        const float cell = 2.0f;  // cells of 0.5 m, +-16 km around the origin of the scene
        int xi = (int)floorf((q.x + o.x) * cell) + 32768, yi = (int)floorf((q.y + o.y) * cell) + 32768;
        xi = xi < 0 ? 0 : (xi > 65535 ? 65535 : xi);
        yi = yi < 0 ? 0 : (yi > 65535 ? 65535 : yi);
        key = hilbert16((uint32_t)xi, (uint32_t)yi);
        // several parameter sets (at most 16 in this mode): the set leads the key, so that every set is one run of the
        // sorted order - a launch of the culling kernel per run, each with its set's constants (csf_engine.hip); the
        // curve index loses its four lowest bits (cells of 2 m instead of 0.5 m)
        if (d.seg_keys) key = ((uint32_t)d.cls[a] << 28) | (key >> 4);
        if (key == 0xFFFFFFFFu) key = 0xFFFFFFFEu;
    }
    keys[a] = key;
    vals[a] = (int32_t)a;
}

__global__ void identity_perm_kernel(const Dev d) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a < d.n_pad) d.perm[a] = (int32_t)a;
}

// the exchange record of a slot of another rank's block -> rec / reclo / rec2 (csf_dev.h: xbuf); returns the record
__device__ __forceinline__ float4 unpack_exchange(const Dev &d, int32_t a) {
    const float4 q = d.xbuf[2 * (int64_t)a], w = d.xbuf[2 * (int64_t)a + 1];
    d.rec[a] = q;
    d.reclo[a] = make_float2(w.x, w.y);
    if (d.has_bike) d.rec2[a] = make_float2(w.z, w.w);
    return q;
}

__global__ void unpack_exchange_kernel(const Dev d) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a < d.n_pad && (a < d.lo || a >= d.hi)) (void)unpack_exchange(d, (int32_t)a);
}

template <bool UNPACK>
__global__ void sorted_copy_kernel(const Dev d) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.n_pad) return;
    const int32_t a = d.perm[p];
    if (a < d.n_pad) d.pos[a] = (int32_t)p;
    // (the padding slot of the class-segmented order sits in many places at once: it is unpacked as often, to the same effect)
    float4 q = (UNPACK && a < d.n_pad && (a < d.lo || a >= d.hi)) ? unpack_exchange(d, a) : d.rec[a];
    if (rec_is_real(q)) {                                     // scene coordinates: offset + the slot's origin
        const float2 o = d.rorg[a];
        const float2 bo = d.borg[p >> 6];                     // (the batches keep their origins between re-binnings)
        d.recb[p] = make_float4(q.x + (o.x - bo.x), q.y + (o.y - bo.y), q.z, q.w);
        q.x += o.x, q.y += o.y;
        d.recg[a] = q;                                        // (a receiver must coincide with itself as a source)
    } else {
        d.recb[p] = q;
    }
    d.recs[p] = q;
    if (d.has_bike) d.recs2[p] = d.rec2[a];
}

// After a re-sort (perm is new, pos / recs / org / bnd are not yet), one wave per batch of 64 places:
//   * every record's new origin: its scene coordinates rounded to 1/4 m, and the record re-expressed as an offset from
//     it - recomputed from the fp64 state where this device holds it for every slot (a single device; any rank right
//     after an upload), else from the old record and the old origin (the foreign blocks of a rank: (old origin - new
//     origin) is exact, the sum rounds once);
//   * pos[], the copy of the records in binned order and scene coordinates, the batch's bounding circle.
// Sentinel records stay as they are (the padding slot of the class-segmented order sits in many places at once).
__global__ __launch_bounds__(256) void rebase_kernel(const Dev d) {
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b * 64 >= d.n_pad) return;
    const int lane = threadIdx.x & 63;
    const int64_t p = b * 64 + lane;
    const int32_t a = d.perm[p];
    float4 q = d.rec[a];
    const bool real = rec_is_real(q);
    const float2 o = real ? d.rorg[a] : make_float2(0.f, 0.f);
    const float x = q.x + o.x, y = q.y + o.y;                  // scene coordinates, rounded: good enough for the box
    float x0 = real ? x : 3e38f, x1 = real ? x : -3e38f, y0 = real ? y : 3e38f, y1 = real ? y : -3e38f;
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, o2, 64));
        x1 = fmaxf(x1, __shfl_xor(x1, o2, 64));
        y0 = fminf(y0, __shfl_xor(y0, o2, 64));
        y1 = fmaxf(y1, __shfl_xor(y1, o2, 64));
    }
    float2 on = make_float2(0.f, 0.f);
    if (real) on = make_float2(0.25f * rintf(4.0f * x), 0.25f * rintf(4.0f * y));
    if (real) {
        float2 lo;
        if (d.rebase_from_state) {
            const double px = (d.s[a] - d.ox) - (double)on.x, py = (d.s[d.cap + a] - d.oy) - (double)on.y;
            q.x = (float)px;
            q.y = (float)py;
            lo = make_float2((float)(px - (double)q.x), (float)(py - (double)q.y));
        } else {   // (the rounding of the sum goes to the low part: the position a rank holds of a foreign source stays what it was)
            const float2 lo0 = d.reclo[a];
            float ex, ey;
            two_sum(o.x - on.x, q.x, q.x, ex);
            two_sum(o.y - on.y, q.y, q.y, ey);
            lo = make_float2(ex + lo0.x, ey + lo0.y);
        }
        d.rec[a] = q;
        d.reclo[a] = lo;
        d.rorg[a] = on;
        d.pos[a] = (int32_t)p;
    } else if (a < d.n_pad) {
        d.pos[a] = (int32_t)p;                                 // (a free slot's place: where an arrival spawned into it will sit)
    }
    {   // the record by place relative to the batch's origin: the new origin of its first road user (0 if there is none)
        const unsigned long long m = __ballot(real);
        const int first = m ? __builtin_ctzll(m) : 0;
        const float fx = __shfl(on.x, first, 64), fy = __shfl(on.y, first, 64);
        d.recb[p] = real ? make_float4(q.x + (on.x - fx), q.y + (on.y - fy), q.z, q.w) : q;
        if (lane == 0) d.borg[b] = make_float2(fx, fy);
    }
    // the binned copy holds scene coordinates, offset + origin (what the tiles of the pair kernels are filled with)
    if (real) q.x += on.x, q.y += on.y;
    d.recs[p] = q;
    if (real) d.recg[a] = q;
    if (d.has_bike) d.recs2[p] = d.rec2[a];
    if (lane == 0) d.bnd[b] = box_circle(x0, x1, y0, y1, 0.0f);
}

__global__ __launch_bounds__(256) void bounds_kernel(const Dev d) {
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b * 64 >= d.n_pad) return;
    batch_circle(d, b, threadIdx.x & 63, 0.0f, d.bnd);
}

// ---- candidate tiles of every receiver group (csf_dev.h: Dev::clist) ----------------------------------------------------
// the circle around the batch circles of one tile (scene coordinates), one wave per tile
__global__ __launch_bounds__(256) void tile_circle_kernel(const Dev d, float4 *tcirc) {
    const int bpt = d.clist_tile / 64;                              // batches per tile: 16 or 32
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t ntiles = (d.n_src + d.clist_tile - 1) / d.clist_tile;
    if (t >= ntiles) return;
    const int lane = threadIdx.x & 63;
    const int64_t b = t * bpt + lane;
    const bool have = lane < bpt && b * 64 < d.n_src;
    const float4 c = have ? d.bnd[b] : make_float4(0.f, 0.f, -1.f, 0.f);
    const bool real = have && fabsf(c.x) < 1e14f;                    // (a batch of sentinels sits at 1e15)
    float x0 = real ? c.x - c.z : 3e38f, x1 = real ? c.x + c.z : -3e38f, y0 = real ? c.y - c.z : 3e38f, y1 = real ? c.y + c.z : -3e38f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, o, 64));
        x1 = fmaxf(x1, __shfl_xor(x1, o, 64));
        y0 = fminf(y0, __shfl_xor(y0, o, 64));
        y1 = fmaxf(y1, __shfl_xor(y1, o, 64));
    }
    if (lane == 0) tcirc[t] = box_circle(x0, x1, y0, y1, 0.0f);
}

// one wave per group of clist_rpb receivers (places of the binned order; a rank's own receivers through rlist): the tiles
// whose circle comes within `reach` of the group's circle, in ascending order
// far (optional, [2]): the largest number of sources any group can meet - the places of its listed tiles and of the tiles that
// are always visited - and the largest sum, over the tiles a group does NOT list, of (places) x exp(-kappa x separation),
// separation = the distance left between the two circles after `move` metres of motion on both sides: what those tiles can add
// to a receiver of the group in units of f_0, whatever happens until the next re-binning (csf_engine.hip: tighten_far_bound).
// Both as atomic maxima over the groups (the sum as the bits of a non-negative float).
__global__ __launch_bounds__(256) void clist_kernel(const Dev d, const float4 *tcirc, uint16_t *clist, int32_t *ccount, float reach,
                                                    unsigned *far, float kappa, float move) {
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nloc = d.hi - d.lo, groups = (nloc + d.clist_rpb - 1) / d.clist_rpb;
    if (g >= groups) return;
    const int lane = threadIdx.x & 63;
    const int64_t j = g * d.clist_rpb + (lane % d.clist_rpb);
    const int64_t jc = j < nloc ? j : nloc - 1;
    const int64_t p = d.rlist ? (int64_t)d.rlist[jc] : d.lo + jc;
    const float4 q = d.recs[p];                                     // scene coordinates by place
    const bool real = rec_is_real(q);
    float x0 = real ? q.x : 3e38f, x1 = real ? q.x : -3e38f, y0 = real ? q.y : 3e38f, y1 = real ? q.y : -3e38f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, o, 64));
        x1 = fmaxf(x1, __shfl_xor(x1, o, 64));
        y0 = fminf(y0, __shfl_xor(y0, o, 64));
        y1 = fmaxf(y1, __shfl_xor(y1, o, 64));
    }
    const float4 gc = box_circle(x0, x1, y0, y1, 0.0f);
    const int64_t ntiles = (int64_t)d.ctail;                         // (tiles from ctail on are always visited)
    int count = 0;
    float tail = 0.0f;
    for (int64_t t0 = 0; t0 < ntiles; t0 += 64) {
        const int64_t t = t0 + lane;
        bool near = false;
        if (t < ntiles) {
            const float4 tc = tcirc[t];
            const float ex = tc.x - gc.x, ey = tc.y - gc.y, rr = reach + tc.z + gc.z;
            const float d2 = ex * ex + ey * ey;
            near = d2 <= rr * rr;
            if (!near && far != nullptr && fabsf(tc.x) < 1e14f)     // (a tile of nothing but sentinels adds nothing)
                tail += (float)d.clist_tile * __expf(-kappa * fmaxf(sqrtf(d2) * 0.9999f - tc.z - gc.z - 2.0f * move, 0.0f));   // (BOTH sides move until the next re-binning, as in `reach`)
        }
        const unsigned long long m = __ballot(near);
        if (near) {
            const int at = count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (at < CLIST_MAX) clist[g * CLIST_MAX + at] = (uint16_t)t;
        }
        count += __builtin_popcountll(m);
    }
    if (lane == 0) ccount[g] = count <= CLIST_MAX ? count : -1;
    if (far != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tail += __shfl_xor(tail, o, 64);
        if (lane == 0) {
            const int64_t all = (d.n_src + d.clist_tile - 1) / d.clist_tile;
            const int64_t met = count <= CLIST_MAX ? (int64_t)count + (all - ntiles) : all;   // (no list: the group walks every tile)
            atomicMax(&far[0], (unsigned)min(met * d.clist_tile, (int64_t)0x7fffffff));
            atomicMax(&far[1], __float_as_uint(count <= CLIST_MAX ? tail * 1.001f : 0.0f));
        }
    }
}

void launch_candidate_lists(const Dev &d, float4 *tcirc, uint16_t *clist, int32_t *ccount, float reach, hipStream_t st, unsigned *far, float kappa,
                            float move) {
    const int64_t ntiles = (d.n_src + d.clist_tile - 1) / d.clist_tile;
    const int64_t nloc = d.hi - d.lo, groups = (nloc + d.clist_rpb - 1) / d.clist_rpb;
    if (ntiles <= 0 || groups <= 0) return;
    hipLaunchKernelGGL(tile_circle_kernel, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, st, d, tcirc);
    hipLaunchKernelGGL(clist_kernel, dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, st, d, tcirc, clist, ccount, reach, far, kappa, move);
}

size_t bin_temp_bytes(int64_t n_pad) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                       (const int32_t *)nullptr, (int32_t *)nullptr, (int)n_pad);
    return bytes;
}

int launch_rebin(const Dev &d, uint32_t *keys, uint32_t *keys_out, int32_t *vals, void *tmp, size_t tmp_bytes,
                 hipStream_t st) {
    if (d.n_pad <= 0) return 0;
    hipLaunchKernelGGL(keys_kernel, dim3((unsigned)((d.n_pad + 255) / 256)), dim3(256), 0, st, d, keys, vals);
    return (int)hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys_out, vals, d.perm, (int)d.n_pad, 0, 32, st);
}

__global__ void receiver_keys_kernel(const Dev d, uint32_t *keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.hi - d.lo) keys[i] = (uint32_t)d.pos[d.lo + i];
}

int launch_receiver_list(const Dev &d, uint32_t *keys, int32_t *rlist_out, void *tmp, size_t tmp_bytes, hipStream_t st) {
    const int64_t m = d.hi - d.lo;
    if (m <= 0) return 0;
    hipLaunchKernelGGL(receiver_keys_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, d, keys);
    return (int)hipcub::DeviceRadixSort::SortKeys(tmp, tmp_bytes, keys, (uint32_t *)rlist_out, (int)m, 0, 32, st);
}

// the sorted slots -> perm with every set's run starting at a multiple of 64 places, sentinel slot in between and behind
__global__ void segment_fill_kernel(const Dev d, int32_t sent_slot) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < d.n_pad) d.perm[p] = sent_slot;
}
__global__ void segment_perm_kernel(const Dev d, const int32_t *sorted_slots, const SegTable tab) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= tab.sorted_beg[tab.n]) return;
    int c = 0;
    while (c + 1 < tab.n && p >= tab.sorted_beg[c + 1]) c++;
    d.perm[tab.place_beg[c] + (p - tab.sorted_beg[c])] = sorted_slots[p];
}

void launch_segment_perm(const Dev &d, const int32_t *sorted_slots, const SegTable &tab, hipStream_t st) {
    if (d.n_pad <= 0) return;
    hipLaunchKernelGGL(segment_fill_kernel, dim3((unsigned)((d.n_pad + 255) / 256)), dim3(256), 0, st, d, tab.sent_slot);
    const int64_t m = tab.sorted_beg[tab.n];
    if (m > 0) hipLaunchKernelGGL(segment_perm_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, d, sorted_slots, tab);
}

// slot -> place and the circles of the batches, for the host (csf_engine.hip: holes_request): written by lanes, straight into mapped
// host memory - no copy command in the stream
__global__ void export_places_kernel(const int32_t *__restrict__ pos, const float4 *__restrict__ bnd, int32_t *__restrict__ hpos,
                                     float4 *__restrict__ hbnd, int64_t np, int64_t nb) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < np) hpos[i] = pos[i];
    if (i < nb) hbnd[i] = bnd[i];
}

void launch_export_places(const Dev &d, int32_t *hpos, float4 *hbnd, int64_t np, int64_t nb, hipStream_t st) {
    if (np <= 0) return;
    hipLaunchKernelGGL(export_places_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, d.pos, d.bnd, hpos, hbnd, np, nb);
}

void launch_identity_perm(const Dev &d, hipStream_t st) {
    if (d.n_pad <= 0) return;
    hipLaunchKernelGGL(identity_perm_kernel, dim3((unsigned)((d.n_pad + 255) / 256)), dim3(256), 0, st, d);
}

void launch_sorted_copy(const Dev &d, hipStream_t st, bool from_exchange) {
    if (d.n_pad <= 0) return;
    if (from_exchange && d.xbuf != nullptr) hipLaunchKernelGGL(sorted_copy_kernel<true>, dim3((unsigned)((d.n_pad + 255) / 256)), dim3(256), 0, st, d);
    else hipLaunchKernelGGL(sorted_copy_kernel<false>, dim3((unsigned)((d.n_pad + 255) / 256)), dim3(256), 0, st, d);
}

void launch_unpack_exchange(const Dev &d, hipStream_t st) {
    if (d.n_pad <= 0 || d.xbuf == nullptr) return;
    hipLaunchKernelGGL(unpack_exchange_kernel, dim3((unsigned)((d.n_pad + 255) / 256)), dim3(256), 0, st, d);
}

void launch_rebase(const Dev &d, hipStream_t st) {
    if (d.n_pad <= 0) return;
    const int64_t batches = d.n_pad / 64;
    hipLaunchKernelGGL(rebase_kernel, dim3((unsigned)((batches + 3) / 4)), dim3(256), 0, st, d);
}

void launch_bounds(const Dev &d, hipStream_t st) {
    if (d.n_pad <= 0) return;
    const int64_t batches = d.n_pad / 64;
    hipLaunchKernelGGL(bounds_kernel, dim3((unsigned)((batches + 3) / 4)), dim3(256), 0, st, d);
}

}  // namespace csf
