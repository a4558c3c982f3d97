// csf_dev.h — device-side view of one engine: SoA agent state in HBM, shared by the pair kernel
// (csf_pair.hip), the per-agent kernel (csf_agent.hip) and the host engine (csf_engine.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/csf.h"

namespace csf {

constexpr int WAVE = 64;
constexpr int MAX_SPLIT = 64;  // max source chunks (partial sums per receiver)

// fp32 constants of the pair kernel, derived on the host from csf_params.
struct PairConsts {
    float sg0, sg1, sg2, sg3, e0, e1;  // vehicle.py:1604-1612
    float lf0;                         // log2(f_0); Bicycle field: log2(p_0 / p_decay) (vehicle.py:1101, 1132)
    float kexp;                        // log2(e)
    float chs;                         // -cos^2(hfov/2) for hfov <= pi, +cos^2 beyond (intersection.py:733-736)
    float ch;                          // cos(hfov/2) (batch classification)
    float chm, chp;                    // ch -+ 1e-4: the margins of that classification (kernel arguments, not registers: csf_pair.hip classify_batch)
    float ipd;                         // Bicycle field: 1 / p_decay (vehicle.py:1095-1099)
    int32_t p2r;                       // intersection.py:739-741
    int32_t f0_zero;                   // vehicle.py:1592-1593
    int32_t fov_classify;              // whole batches are classified against the field-of-view cone
    int32_t full_circle;               // hfov >= 2 pi: every bearing is inside the field of view
    float rfar;                        // sources farther than this add less than far_eps * f_0 / n in magnitude (inf: off)
    // per-pair reach test of the cull-first kernel (csf_pair.hip: keep_x2): T (sigma_a - sigma_b / 2) and T sigma_b / 2 as
    // polynomials in s2 = sin^2(psi0 - psi), T = ln(n / far_eps) with a margin for fp32 rounding
    float tA0, tA1, tB0, tB1;
    int32_t reach;                     // the reach test is on (far-field cull enabled)
    // pairs closer than rnear are evaluated from the precise records (csf_pair.hip: precise_delta); rnear2 = rnear^2,
    // rn2big = 1e20 rnear^2 (what makes the fast path drop such a pair without a branch)
    float rnear, rnear2, rn2big;
    // Field-of-view decisions within fp32 rounding of an edge (csf_field.h: keep_x2 / tracked_m; csf_engine.hip: set_fov_band).
    // With t = rho cos(bearing) the fast test is g = t|t| + chs rho^2 > 0; positions in the tiles are off by eps_p (2^-24 of the
    // largest coordinate), headings by 2^-24, so |g - exact| < 12 eps_p rho + 18 2^-24 rho^2.  A pair with |g| below twice that
    // - fovA r2 + fovB after rho <= r2 / 16 + 4 - is MARGINAL: the fast path keeps it, and it is decided as the reference
    // decides (fp64 atan2 -> limitAngle -> angleDifference, intersection.py:711-736) where it is corrected.  sideA / sideB:
    // the same band for the side test of priority-to-the-right (:739-741), |rho sin(bearing)| < sideA r2 + sideB.
    float fovA, fovB, sideA, sideB;
    // the packed test of the cull-first kernel (keep_x2) works on the cosine of the bearing, t / rho against chk = cos(hfov/2)
    // (-2 for a full circle), band fovT0 + fovT1 / rho = 2 (8 u + 3 eps_p / rho)
    float chk, fovT0, fovT1;
    float clsk;                        // classify_batch: the receiver must be outside a batch's circle by clsk x (their coordinates): the
                                       // position error of fp32, 2^-24 of a coordinate, as an angle below half the 1e-4 margin
    // the same band for a pair formed from the PRECISE records (csf_pair.hip: precise_delta; offsets of a few metres from
    // origins whose difference is exact): |g| < fovP1 rho + fovP2 r2, |rho sin(bearing)| < sideP0 + sideP1 rho
    float fovP1, fovP2, sideP0, sideP1;
};

// A pair whose field-of-view decision is within rounding even on the precise records (about one in 1e7): the pair kernel
// hands it to the per-agent kernel, which decides it as the reference does - fp64, atan2 -> limitAngle -> angleDifference
// (intersection.py:711-741; csf_dev.h: untracked_exact_xy) - on the receiver's own fp64 state and the source's position
// as it was when the pair kernel ran, and adds or removes the pair's force (csf_agent.hip: COMBINE).
// The same for np.sign(phi) of the TwoD field (vehicle.py:1625): where the receiver sits within rounding of the line ahead of
// the source the pair kernel evaluates the pair with sign +1 (fx, fy) and with sign -1 (fx2, fy2), and the per-agent kernel
// picks by the reference's own chain - acos -> limitAngle -> sign (csf_dev.h: sign_phi_exact).
struct EdgeRec {
    double xi, yi;        // the source's position (fp64 state at the time of the pair launch)
    double psi;           // ... its heading (EDGE_SIDE entries; EDGE_HEADING_REC: (cos, sin) of the record, low and high word)
    double hfov;          // ... and its field of view (intersection.py:733-735: the source's parameter set)
    float fx, fy;         // the force of the pair, from the precise records (EDGE_SIDE: with np.sign(phi) = +1)
    float fx2, fy2;       // EDGE_SIDE: the force with np.sign(phi) = -1
    int32_t recv;         // the receiver's slot
    int32_t next;         // 1 + ring index of the next entry of this receiver, 0: none
    uint32_t stamp;       // Dev::edge_stamp of the launch that wrote it
    int32_t flags;        // EDGE_SEEN: the pair kernel took the source for tracked (and added fx, fy); EDGE_SIDE
};
constexpr int32_t EDGE_SEEN = 1, EDGE_SIDE = 2, EDGE_HEADING_REC = 4;   // HEADING_REC: psi holds the record's (cos, sin) as two floats
constexpr unsigned EDGE_CAP = 4096;
constexpr uint32_t CSF_ST_EDGE = 0x80000000u;   // status[slot], internal: entries wait in the ring for this receiver (masked by csf_status)   // ring of entries; a tick produces a few dozen at N = 16 384

// one run of the class-segmented order (csf_engine.hip: rebin), as the segmented grid of the pair kernel reads it
struct SegDev {
    PairConsts pc;           // the set's constants (the rounding bands of the tick come from the launch's own Dev::pc)
    double hfov;
    int64_t src_beg, n_src;  // its places
    int32_t chunk_units, n_split, part_base;
    int32_t first_by;        // its first source chunk in the grid's y dimension
};

constexpr int STATE_ROWS = 8;   // rows of the state array: the widest vehicle.s (BalancingRiderBicycle, vehicle.py:1959-1960)

// All arrays have `cap` elements per component unless noted; component c of agent a is at [c*cap + a].
struct Dev {
    csf_params p;
    PairConsts pc;
    // PlanarBicycle (csf_engine.hip: derive_planarbike): one exact step of z' = M z + (1/G, 0)^T u for z = (v delta / l, psi),
    // M = [[p1 + p2, -p1 p2], [1, 0]]: z+ = E z + G u, the same for every speed.  pb = (E[4] row-major, G[2], exp(-k_p_v t_s))
    double pb[7];
    // Parameter sets (csf_set_param_classes; every reference vehicle owns its params object, vehicle.py:64-204): the table
    // (n_classes >= 1 rows; row 0 == p), what derive_consts makes of each row, and the row of every slot.  With one row
    // the kernels read p / pc / pb from their arguments; with more, the per-agent kernel and the pair kernel read the rows.
    int32_t n_classes;
    int32_t model_mask;      // bit m: some parameter set is of vehicle class m (enum csf_model); several bits: a mixed population
    int32_t has_bike;        // ... bit CSF_BICYCLE: the second record (rec2: e, 1/sqrt(1-e^2) of the Bicycle field) is maintained
    const csf_params *ptab;
    const PairConsts *pctab;
    const double *pbtab;     // [n_classes][7]
    uint8_t *cls;            // [cap]
    int64_t n;         // agent SLOTS in use (the highest one + 1); a slot may be dead after csf_remove_agents until it is reused
    int64_t n_live;    // road users (intersection.py n_bikes)
    int64_t cap;       // SoA stride
    int64_t lo, hi;    // receiver block integrated by this rank
    int64_t n_pad;     // n rounded up to a multiple of 64 (sentinel source records behind n)
    int64_t n_src;     // places of the source order that can hold a road user (multiple of 64, <= n_pad): the pair kernel stops there
    int64_t src_beg;   // ... and starts here (0, or the first place of a class segment: csf_engine.hip launch_pair_segments)
    int32_t part_base; // first slot of d.part this launch writes (its source chunks follow: one launch per class segment)
    const SegDev *segtab;  // the runs whose field is the TwoD one, for the segmented grid (NULL: one launch per run)
    int32_t n_seg;
    int32_t seg_keys;  // re-binning: the parameter set leads the sort key (csf_bin.hip), so that every set is a run of places
    int32_t ns;        // states per agent
    int32_t hist_len;  // power of two > int(1/t_s) + 1: short on-device position ring
    int32_t back;      // int(1 / t_s) — vehicle.py:1487
    int32_t n_split;   // source chunks of the pair kernel (every one of them holds sources: csf_engine.hip set_shard)
    int32_t chunk_units;  // batches of 64 records per source chunk
    int32_t dyn_recv;  // cull kernel: receivers handed to the waves of a workgroup dynamically
    int32_t rpb;       // ... and receivers per workgroup then: 16 or 32
    int32_t wide;      // rpb 32, receivers in slot order: workgroups of 8 waves on tiles of 2048 sources (chunk_units is then 32)
    int32_t pair_variant;  // 0: cull-first kernel on binned records (default), 1: evaluate-then-mask, 2: cull-first, unbinned
    int32_t classify;      // the records are binned and every batch of 64 carries a bounding circle
    double ox, oy;     // origin of the scene: what the batch origins (org) and the bounding circles (bnd) are relative to
    int64_t tick;

    double *s;         // [STATE_ROWS][cap]  x, y, psi, v, delta, theta (roll), + BalancingRider: steer rate, roll rate
    double *vdes;      // [cap]
    int64_t *qbeg;     // [cap] first row of the agent's destination queue in q
    int32_t *qlen;     // [cap] rows of it
    double *q;         // [qcap][3] rows (x, y, stop): a slab, every queue contiguous, new and replaced queues appended
    int64_t qcap;
    uint8_t *alive;    // [cap] 0: dead slot (sentinel record, not integrated)
    const int32_t *order;  // [n_live] road user i of the host's population order -> slot
    int32_t *ptr;      // destpointer
    uint8_t *znav;     // 0 cruise, 1 brake, 2 arrived; bit 7: not one-hot
    double *znp;       // [3][cap] latched v0, d0, d1 (vehicle.py:428-430)
    int32_t *ti;       // column of the reference's traj ring (vehicle.py:1279-1280)
    double *hx, *hy;   // [hist_len][cap]
    double *lti;       // [5][cap] InvPendulum LTI state (vehicle.py:1728)
    uint8_t *zrid;     // 1 riding, 0 walking (vehicle.py:1732-1736)
    int32_t *dgood;    // consecutive ring samples with |delta| < delta_max_walk (vehicle.py:1943-1947)
    double *ppsi;      // PlanarPoint unwrapped yaw (dynamics.py:943-966)
    // UncontrolledVehicle (vehicle.py:920-988): the prescribed trajectory of a slot, rows (x, y, psi, v) of a slab
    const double *script;
    const int64_t *sbeg;   // [cap] first row
    int32_t *slen;         // [cap] rows (0: none given - the reference's ring of zeros behind the start state)

    // The PRECISE fp32 source records (x - ox - rorg.x, y - oy - rorg.y, cos psi, sin psi): the position is an offset from
    // the road user's OWN origin - where it was at the last re-binning, rounded to 1/4 m (exact in fp32, and so is the
    // difference of two origins) - not from the scene origin.  The offset stays below a few metres (0.07 m per tick, 32
    // ticks), so a position resolves to ~2e-7 m whatever the extent of the scene.  This is what is all-gathered, what
    // the re-binning reads, and what near pairs are evaluated from (csf_pair.hip: precise_delta); the tiles of the pair
    // kernels hold scene coordinates (recs, or offset + origin), 2^-24 of the scene extent.
    float4 *rec;       // [n_pad] by slot
    float2 *rorg;      // [n_pad] by slot: the origin its record is relative to, itself relative to (ox, oy)
    int32_t keep_lo;   // this engine is a rank of a sharded run (or a member of a loopback group): reclo is maintained
    float2 *reclo;     // [n_pad] by slot: what the record's position left over, (offset - fp32(offset)) in fp32: origin +
                       // record + this is the fp64 position to ~1e-14 m.  Exchanged with the records, so that a rank can
                       // hand a foreign source's position to the exact field-of-view decision (csf_field.h: edge_handover)
    // A rank of a sharded run (or a member of a loopback group): what travels is ONE array of exchange records, 32 bytes per
    // slot - (record) and (low part, second record) -, so that a tick is one collective on one contiguous block per rank
    // instead of a group of two or three.  The per-agent kernel writes its road user's entry beside the arrays above
    // (write_record); what arrives for the other ranks' blocks is spread to rec / reclo / rec2 by the copy into binned order
    // that a rank runs anyway (csf_bin.hip: sorted_copy_kernel), or by a launch of its own where that does not run.
    float4 *xbuf;      // [n_pad][2], NULL on an unsharded engine
    // Large populations (recv_binned; csf_pair.hip BINR works relative to the origin of its receiver group): the record by
    // place of the binned order, its position relative to the origin of its BATCH of 64 places - the own origin of the
    // batch's first road user at the last re-binning - formed as offset + (own origin - batch origin), the bracket exact:
    // one rounding at the extent of a batch (2e-6 m), whatever the extent of the scene.
    float4 *recb;      // [n_pad] by place: (x, y relative to borg[place / 64], cos psi, sin psi)
    float2 *borg;      // [n_pad / 64] those origins, relative to (ox, oy); (0, 0) for a batch without a road user
    float4 *recg;      // [cap] by slot: the record in scene coordinates, bit for bit what recs holds at the slot's place (the
                       // receivers of the kernels on binned records: a receiver must coincide with itself as a source)
    int32_t state_current;       // pair kernels: the fp64 state d.s of EVERY live slot is current on this device (not on a rank of a
                                 // sharded run once it has ticked): marginal field-of-view decisions are then taken from it
    EdgeRec *edge;               // [EDGE_CAP] ring (NULL: the pair launch hands nothing over - shards, csf_count_pairs)
    unsigned *edge_n;            // [1] entries appended since the engine was created (ring index = count % EDGE_CAP)
    int32_t *edge_head;          // [cap] 1 + ring index of the newest entry of the receiver in this slot, 0: none; valid while
                                 // bit 31 of status[slot] is set (CSF_ST_EDGE: the per-agent kernel loads the status word anyway)
    uint32_t edge_stamp;         // names the pair launch(es) of this tick; entries of another stamp are stale
    unsigned *near_dropped;      // [1] near / marginal pairs a full per-wave list could not take (csf_pair.hip: near_note); never reset
    int32_t rebase_from_state;   // re-binning: every live slot's fp64 state is current on this device (else: re-express the old record)
    float2 *rec2;      // [n_pad] Bicycle field only: (e, 1/sqrt(1-e^2))
    int32_t *perm;     // [n_pad] spatially binned order of the source records (position -> record index)
    int32_t *pos;      // [n_pad] inverse of perm (record index -> position)
    float4 *recs;      // [n_pad] the records in binned order AND scene coordinates (x - ox, y - oy, cos psi, sin psi)
                       // (single device: written by the agent kernel beside rec, so that the pair kernel's tile
                       // fill is one coalesced load instead of perm -> rec, rorg)
    float2 *recs2;     // [n_pad] Bicycle field: rec2 in binned order
    int32_t recs_valid;
    int32_t recv_binned;   // the pair kernel takes its receivers in binned order too and skips far tiles (large populations)
    const int32_t *rlist;  // recv_binned on a shard: binned positions of this rank's receivers, ascending (NULL: all of them)
    // Receivers in binned order (recv_binned): which tiles can matter to a receiver group until the next re-binning.  At every
    // re-binning one wave per group tests the circle of every tile (csf_bin.hip: clist_kernel) and keeps those within the
    // far-field radius + what both sides can move in the meantime; the pair kernel then visits the tiles of its group's list
    // (each still tested against this tick's circles) instead of walking all of them - O(N / tile) tests per group and tick
    // at 262 144 or a million road users.  Tiles from ctail on (the sentinel tail, where arrivals appear) are always visited.
    const uint16_t *clist;     // [groups][CLIST_MAX] tile numbers (tile = clist_tile sources of the binned order), ascending; NULL: none
    const int32_t *ccount;     // [groups] entries, or -1: more than CLIST_MAX candidates - this group walks every tile
    int32_t clist_tile;        // sources per tile the lists were built for (1024 / 2048), receivers per group (= rpb then)
    int32_t clist_rpb;
    int32_t ctail;             // first tile that is always visited
    float4 *bnd;       // [n_pad/64] bounding circle (cx, cy, radius, -) of every batch of 64 binned records
    float4 *bnd_next;  // written by the pair kernel for the next tick (from this tick's records + bnd_margin)
    float bnd_margin;  // largest distance an agent can move in one tick (t_s * v_max)
    float2 *part;      // [MAX_SPLIT][cap] partial repulsive sums of the pair kernel
    float2 *froad;     // [cap]
    float4 *rv;        // [nv_pad] road vertices (x - ox - rvo.x, y - oy - rvo.y, -F0, -(sigma+1)/2): offsets from the origin of
                       // their tile of 1024 vertices (a multiple of 1/4 m), so that a vertex resolves to 2^-24 of a tile's extent
    float2 *rvo;       // [nv_pad / 1024 + 1] those origins, relative to (ox, oy)
    int64_t nv, nv_pad;
    int32_t road_np;   // sigma + 1 when every road edge shares one integer sigma in 1..5, else 0 (road_kernel)
    // Large road networks (csf_road.hip; csf_engine.hip: build_road_grid): the vertices are static, so the road term is split
    // over a lattice of square cells - a receiver sums the vertices of the 5 x 5 cells around its own directly and takes the
    // rest of the network from the cell's Chebyshev interpolant of that (smooth) far field, fitted once per network.
    int32_t rg_nx, rg_ny;      // cells of the lattice (0: no lattice, road_kernel sums every vertex)
    int32_t rg_by_place;       // receivers are taken in binned order (neighbours share their cells' vertices in cache)
    float rg_w;                // edge of a cell
    float rg_x0, rg_y0;        // corner of cell (0, 0) relative to (ox, oy): a multiple of rg_w
    const float4 *rg_v;        // [nv] vertices sorted by cell: (x, y) relative to the CENTRE of their cell, -F0, -(sigma+1)/2
    const int32_t *rg_start;   // [nx ny + 1] first vertex of every cell (row-major)
    const float *rg_c;         // [nx ny][2][64] coefficients c_ab of sum_ab c_ab T_a(xi) T_b(eta), x and y component

    double *F;         // [F_rows][cap] Fx, Fy, Fdest_x, Fdest_y, Frep_x, Frep_y
    int32_t F_rows;    // 6; 2 in the view of csf_replay_forces (Fx, Fy of one recorded tick): loads clamp their row to F_rows - 1
    uint32_t *status;

    // The fused tick of mid-size populations (csf_mid.hip: one launch = pair sums + per-agent tick): a receiver group's per-agent
    // phase runs as soon as ITS sums are complete, while other groups still read this tick's records - so the next tick's records
    // go to the other half of a double buffer, and the fp64 positions an undecidable pair is handed over with (csf_field.h:
    // edge_handover) come from a snapshot of the tick's start instead of the state that is being overwritten.
    float4 *rec_w, *recg_w;   // where write_record puts the record: rec, recg themselves, or the other half
    float2 *rec2_w;
    const double *src64;      // [3][cap] (x, y, psi) of every slot as the pair kernels' hand-overs read them: s itself, or the snapshot
    double *src64_w;          // fused tick: the snapshot the per-agent phase leaves for the next tick (NULL otherwise)
    int32_t mid_group;        // fused tick: road users (slots) per workgroup: 4, 8, 16 or 32
    // The per-agent launch INSIDE the pair launch's drain (round 6; csf_engine.hip: enqueue_chase_tick).  At the headline size the
    // pair launch spends its last ~24 us with most CUs idle while the per-agent kernel (8 us + two launch gaps) waits behind it; a
    // group of 64 road users needs only ITS partial sums.  The per-agent kernel runs on the engine's second stream beside the
    // pair launch: a one-wave gate kernel in front of it holds it back until most pair workgroups are through, then every wave
    // runs its destination-force phase at once, polls the tagged granules of its road users' partial sums (part4 below: written
    // through by the pair workgroups, nobody waits, nobody signals - MI355X_MICROARCH.md: granules) and carries on.  Next tick's records go to the other halves of the
    // double buffers the one-launch tick already uses (+ recs_w), the fp64 positions of hand-overs come from the snapshot.
    float4 *recs_w;           // where write_record puts the binned copy: recs itself, or the other half
    float4 *part4;            // [16][cap] the partial sums of such a launch as 16-byte granules (x, y, tag, 0), each written by ONE
                              // write-through store: the per-agent wave polls the granules of its road users until every tag is this
                              // tick's (chase_tag) - the producer neither waits for its stores nor signals.  NULL: an ordinary launch
    uint32_t chase_tag;       // ... this tick's tag: never 0, never repeated while old granules can still be read
    unsigned *chase_misc;     // [0] pair workgroups through since it was cleared (the gate's counter), [1] waits that gave up (an error)
    unsigned *chase_err;      // ... and the word of MAPPED HOST memory a wait that gives up sets to 1: csf_sync reads it without a copy
    uint32_t chase_round;     // ticks since the through-counter was cleared, this one included
    uint32_t chase_gate;      // the gate opens at this many pair workgroups through
    uint32_t chase_slot;               // ... row of this tick in it
    unsigned long long *chase_clock;   // CSF_CHASE_CLOCK (tools/chase_clock.py): [128 ticks][8] wall_clock64 stamps - pair: first start, last end;
                                       // gate: entry, exit; per-agent: first entry, last past its wait, last end - else NULL

    const int32_t *replay_len;  // csf_replay_forces: per-agent number of ticks (NULL = all), and the tick within
    int64_t replay_tick;        // the replay
    double *hist;      // opt-in history [hist_cap][n][ns]
    int32_t hist_stride, hist_cap;
    unsigned long long *pair_count;  // csf_count_pairs: [4] pair evaluations, per-lane tests, full and partial evaluation passes of the launch (NULL: not counted)
    double *snap;      // csf_step_get_tick on the one-wave path: the packed read-back of csf_get_tick, written by the kernel behind its last
                       // tick (a mapped host buffer), else NULL
    uint64_t *atrace;  // CSF_TRACE_AGENT: eight time stamps (wall_clock64, 100 MHz) of every wave of the last per-agent launch, else NULL:
                       // entry, own scalars loaded, destination force done, partial sums loaded, combine done, integrate done, stores
                       // issued, stores done
    uint64_t *trace;   // CSF_TRACE_BLOCKS: (start, end, hw id) of every pair-kernel workgroup of the last tick, else NULL
};

enum : int { PH_DEST = 1, PH_COMBINE = 2, PH_INTEGRATE = 4, PH_FIXSPEED = 8 };

// launchers implemented in csf_pair.hip / csf_agent.hip
// t0 / t1 (may be NULL): HIP events that receive the start / end time stamp of the kernel itself (hipExtLaunchKernelGGL:
// taken from the dispatch packet, no extra barrier packet in the stream as with hipEventRecord)
void launch_pair(const Dev &d, hipStream_t st, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
void launch_pair_segments(const Dev &d, int total_by, hipStream_t st, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
void launch_road(const Dev &d, hipStream_t st, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
// csf_road.hip: the road term over a lattice (Dev::rg_*).  launch_road_far: the far field of every cell at its 8 x 8
// Chebyshev nodes, samples[(cell * 64 + node) * 2 + component] (node = 8 ix + iy), from the cell-sorted vertices and
// their cells (x, y index); launch_road_grid: the per-tick kernel.
constexpr int RG_NEAR = 2;      // cells on either side of a receiver's own whose vertices are summed directly
constexpr int RG_NODES = 8;     // Chebyshev nodes per direction
void launch_road_far(const Dev &d, const short2 *vcell, double *samples, hipStream_t st);
void launch_road_grid(const Dev &d, hipStream_t st, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
void launch_agent(const Dev &d, int phases, hipStream_t st, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
// the whole per-agent tick beside the pair launch that feeds it (Dev::part4), behind its gate; false: not built for this class
bool launch_agent_chase(const Dev &d, hipStream_t st, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
void preload_chase_kernels();
void launch_chase_scratch_warm(unsigned *out, hipStream_t st);
// the other halves of the double buffers <- this tick's, the gate's counter (through: may be NULL) cleared: one launch
void launch_chase_sync(const Dev &d, float4 *rec_o, float4 *recg_o, float4 *recs_o, float2 *rec2_o, double *cur, int64_t nrecg, unsigned *through,
                       hipStream_t st);
constexpr unsigned CHASE_SPIN_LIMIT = 1u << 18;   // polls (each behind an s_sleep: ~0.3 s in all) before a wait gives up - every wave exits
// csf_agent.hip: up to SMALL_MAX road users of one TwoD-field class, n_ticks whole ticks in one launch of one wave (the rounding
// bands of the launch: csf_engine.hip set_fov_band)
constexpr int SMALL_MAX = 32;
constexpr int SMALL_ROAD_MAX = 2048;   // road vertices (padded) the one-wave kernel stages; and at most 256 of them per lane and tick
void launch_small_tick(const Dev &d, int n_ticks, hipStream_t st, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
// csf_mid.hip: one tick of a mid-size population (plain pair sums + per-agent tick) in one launch; d.rec_w / recg_w / rec2_w /
// src64_w point at the halves of the double buffers this tick does not read, d.mid_group = slots per workgroup
bool launch_mid_tick(const Dev &d, hipStream_t st, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
const char *pair_kernel_name(const Dev &d);           // the kernel launch_pair() takes for this engine
void launch_records(const Dev &d, hipStream_t st);  // rebuild fp32 records from the fp64 state
// csf_get_tick: row-major state [n, ns], Fx [n], Fy [n] (doubles), destination pointers [n] (int32), navigation state
// one-hot [n, 3] (bytes), packed behind each other in `out` (host-mapped)
void launch_snapshot(const Dev &d, double *out, hipStream_t st);
// csf_bin.hip: spatial binning of the source records (Hilbert order) and per-batch bounding circles
size_t bin_temp_bytes(int64_t n_pad);
// class-segmented order: the runs of equal parameter set in the sorted order moved to batch-aligned places (csf_bin.hip)
struct SegTable {
    int32_t n;                 // parameter sets
    int32_t sent_slot;         // a slot whose record is a sentinel for ever (fills the padding)
    int64_t sorted_beg[17];    // first sorted position of set c (sets in ascending order; [n] = road users)
    int64_t place_beg[16];     // first place of set c in the padded order (multiple of 64)
};
void launch_segment_perm(const Dev &d, const int32_t *sorted_slots, const SegTable &tab, hipStream_t st);
int launch_rebin(const Dev &d, uint32_t *keys, uint32_t *keys_out, int32_t *vals, void *tmp, size_t tmp_bytes,
                 hipStream_t st);
void launch_identity_perm(const Dev &d, hipStream_t st);
void launch_export_places(const Dev &d, int32_t *hpos, float4 *hbnd, int64_t np, int64_t nb, hipStream_t st);
// pos[] and recs[] from perm[] and rec[]; from_exchange: the records of the other ranks' blocks are taken from the exchange
// records that have just arrived (Dev::xbuf) and written to rec / reclo / rec2 on the way
void launch_sorted_copy(const Dev &d, hipStream_t st, bool from_exchange = false);
void launch_unpack_exchange(const Dev &d, hipStream_t st);   // ... that alone
// after a re-sort: every record re-expressed relative to its new origin (from the fp64 state where
// d.rebase_from_state), pos[] / recs[] and the bounding circles - one launch
void launch_rebase(const Dev &d, hipStream_t st);
void launch_bounds(const Dev &d, hipStream_t st);
// candidate tiles of every receiver group (Dev::clist): tile circles into tcirc [n_pad / tile], then the lists
constexpr int CLIST_MAX = 128;
// far (may be NULL): [2] zeroed words that receive the groups' maxima of (sources a group can meet, what the tiles it does not list
// can add at most, in units of f_0, as float bits) for the decay rate kappa and `move` metres of motion on either side
void launch_candidate_lists(const Dev &d, float4 *tcirc, uint16_t *clist, int32_t *ccount, float reach, hipStream_t st,
                            unsigned *far = nullptr, float kappa = 0.0f, float move = 0.0f);
// binned positions of the receivers [lo, hi) of this rank in ascending order (the rank's receivers as neighbours in space)
int launch_receiver_list(const Dev &d, uint32_t *keys, int32_t *rlist_out, void *tmp, size_t tmp_bytes, hipStream_t st);

// population changes on the device (csf_add_agents / csf_remove_agents / csf_set_dest_queue between ticks, without the
// round trip through the host mirror): see csf_agent.hip
struct SpawnRec {      // one new road user (Vehicle.__init__, vehicle.py:64-204)
    int32_t slot, qlen;
    int64_t qbeg;
    double s[STATE_ROWS], vdes;
    int32_t cls, pad;  // its parameter set (csf_set_agent_class before the batch reaches the device)
};
struct QueueRec {      // one replaced destination queue (Vehicle.setDestinations, vehicle.py:606-647)
    int32_t slot, qlen;
    int64_t qbeg;
    int32_t mode, pad;  // 1: pointer rewinds, 0 / 2: pointer kept (clamped to the new length)
};
// one batch of population changes, laid out in one pinned, device-visible host buffer: the header, then the records
struct PatchHeader {
    int64_t n_retire, n_spawn, n_requeue, n_rows;   // int32 slots | SpawnRec | QueueRec | rows [n_rows][3] for d.q[q_top ..]
    int64_t off_retire, off_spawn, off_requeue, off_rows;   // byte offsets from the header
    int64_t q_top;                                   // first row of the queue slab that the rows go to
};
void launch_patch(const Dev &d, const PatchHeader &h, const void *records, unsigned *ticket, int b0, int b1, int64_t items,
                  hipStream_t st);
void launch_untracked(const Dev &d, uint8_t *out, hipStream_t st);   // get_untracked_foes as the reference's matrix
void launch_nav_kat(const Dev &d, const int32_t *idx, int64_t m, int what, const int32_t *stop, double *vd_out,
                    double *ddest_out, hipStream_t st);
void launch_pair_kat(const Dev &d, const float4 *src, const float2 *src2, const float4 *recv, int64_t m,
                     int apply_fov, float2 *out, hipStream_t st);

// The kernel's arguments (csf_dev.h: Dev, 1.3 KB with the parameter set) live in the kernarg segment, which the host has just
// written: the first scalar load from each of its 64-byte lines goes to memory (~0.4 us), and the compiler loads a member where
// it is first used - a dozen such round trips along the dependent chain of the per-agent kernel, which is ONE wave per CU (58 % of a wave's
// life was spent parked at s_waitcnt: profiles/r4_agent_kernel_pmc.json), seven at the start of the plain pair kernel.  One word of every line is asked for at entry, all
// at once: one round trip, and every later scalar load hits the scalar cache.
template <int BYTES>
__device__ __forceinline__ uint32_t kernarg_touch() {
    const __attribute__((address_space(4))) uint32_t *ka = (const __attribute__((address_space(4))) uint32_t *)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < (BYTES + 63) / 64; i++) acc |= ka[16 * i];
    return acc;
}

// ... and the value has to be used: `kernarg_touched(x)` behind the kernel's first own loads costs nothing.
__device__ __forceinline__ void kernarg_touched(uint32_t lines) { asm volatile("" ::"s"(lines)); }

constexpr double PI = 3.141592653589793238462643383279502884;

// Data that WAVES of one workgroup hand to each other through memory (csf_mid.hip: status bits and hand-over entries that the
// pair waves leave for the wave that runs the per-agent tick): a CU's vector L1 may hold an older copy of the line, and it does
// not see atomics, so the consumer's loads are agent-scope loads (`sc1`: they bypass the L1 and are served by the XCD's L2,
// where the producers' stores and atomics have landed once their `s_waitcnt vmcnt(0)` has passed).  PUB = false: plain accesses.
template <bool PUB, class T>
__device__ __forceinline__ T ld_pub(const T *p) {
    if (PUB) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// 16-byte granules handed from a workgroup to a wave of another kernel within a launch (csf_dev.h: part4): ONE write-through store, ONE
// agent-scope load - data and tag travel together, so no ordering between separate accesses is needed (MI355X_MICROARCH.md: granules)
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_granule16(float4 *p, float x, float y, float z, float w) {
    const v4f_t v = {x, y, z, w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
// eight of them asked for at once, one wait: the values are operands of the wait, so that nothing that uses them moves in front of it
__device__ __forceinline__ void ld_granules16_x8(const float4 *p, int64_t stride, int n, v4f_t (&g)[8]) {
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const float4 *q = p + (int64_t)(c < n ? c : n - 1) * stride;       // (clamped: the duplicates are not looked at)
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(g[c]) : "v"(q) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7])::"memory");
}

// utils.py:124-139
__device__ __forceinline__ double limit_angle(double th) {
    th = floor(th / (2 * PI)) * (-2 * PI) + th;
    if (th > PI) th -= 2 * PI;
    else if (th < -PI) th += 2 * PI;
    return th;
}

// utils.py:167-182: signed shortest rotation a1 -> a2 (ties resolve to +)
__device__ __forceinline__ double angle_diff(double a1, double a2) {
    double da = fabs(a1 - a2);
    if (da > PI) da = 2 * PI - da;
    double t1 = fabs(limit_angle(a1 - da) - a2), t2 = fabs(limit_angle(a1 + da) - a2);
    return t1 < t2 ? -da : da;
}

// intersection.py:711-741 as the reference decides it - fp64, atan2 -> limitAngle -> angleDifference - on the fp64 state
// of both road users: does the receiver in slot aj ignore the source in slot ai (ai != aj)?  hfov: the SOURCE's (:733-735).
__device__ __forceinline__ bool untracked_exact_xy(double xi, double yi, double xj, double yj, double psij, double hfov, bool p2r);
__device__ __forceinline__ bool untracked_exact(const Dev &d, int64_t ai, int64_t aj, double hfov, bool p2r) {
    return untracked_exact_xy(d.s[ai], d.s[d.cap + ai], d.s[aj], d.s[d.cap + aj], d.s[2 * d.cap + aj], hfov, p2r);
}

// the same decision for a source at (xi, yi) and the receiver (xj, yj, psij)
__device__ __forceinline__ bool untracked_exact_xy(double xi, double yi, double xj, double yj, double psij, double hfov, bool p2r) {
    const double az = limit_angle(atan2(yi - yj, xi - xj));        // :711-718
    const double rel = angle_diff(psij, az);                       // :724-726
    return (fabs(rel) > hfov / 2) | (p2r & (rel > 0));             // :733-741
}

// np.sign(phi) of vehicle.py:1617-1625 as the reference computes it - cart2polar (acos, utils.py:185-194), limitAngle - for
// the source (xi, yi, psii) and a receiver at (xj, yj): -1, 0 or +1
__device__ __forceinline__ int sign_phi_exact(double xi, double yi, double psii, double xj, double yj) {
    const double dx = xj - xi, dy = yj - yi;                       // :1615-1616
    const double rho = sqrt(dx * dx + dy * dy);
    double p = acos(dx / rho);
    if (dy < 0) p = -p;
    const double phi = limit_angle(p - psii);                      // :1618
    return (phi > 0) - (phi < 0);
}

// a + b as an unevaluated sum hi + lo (Knuth's TwoSum: no ordering of |a|, |b| assumed)
__device__ __forceinline__ void two_sum(float a, float b, float &hi, float &lo) {
    hi = a + b;
    const float bb = hi - a;
    lo = (a - (hi - bb)) + (b - bb);
}

// sentinel records (padding, holes of a shard, slots left by csf_remove_agents) sit at 1e15 m: they contribute exactly
// nothing as sources (exp2 underflow) and take no part in origins, circles and sort keys
__device__ __forceinline__ bool rec_is_real(const float4 q) { return fabsf(q.x) < 1e14f; }

// circumcircle of a bounding box (scene coordinates), rounded up: it must contain every record of the batch, whose scene
// coordinates are themselves rounded sums origin + offset (2^-24 relative)
__device__ __forceinline__ float4 box_circle(float x0, float x1, float y0, float y1, float margin) {
    if (x1 < x0) return make_float4(1e15f, 1e15f, 0.0f, 0.0f);   // nothing but sentinels: a circle beyond every far-field radius
    const float w = x1 - x0, h = y1 - y0;
    const float cx = 0.5f * (x0 + x1), cy = 0.5f * (y0 + y1);
    const float rad = 0.5f * sqrtf(w * w + h * h) * 1.0001f + 1e-4f + margin + 2.4e-7f * (fabsf(cx) + fabsf(cy) + w + h);
    return make_float4(cx, cy, rad, 0.0f);
}

// bounding circle of batch b (64 records in perm order), computed by one wave: centre (scene coordinates) and radius of
// the bounding box's circumcircle, grown by `margin`
__device__ __forceinline__ void batch_circle(const Dev &d, int64_t b, int lane, float margin, float4 *out, bool pub = false) {
    // scene coordinates of the batch's records: the binned copy, or (without it) offset + origin of the slot
    float4 q;
    if (d.recs_valid) {
        q = d.recs[b * 64 + lane];
    } else {
        const int32_t a = d.perm[b * 64 + lane];
        q = d.rec[a];
        if (rec_is_real(q)) q.x += d.rorg[a].x, q.y += d.rorg[a].y;
    }
    const bool real = rec_is_real(q);
    const float x = q.x, y = q.y;
    float x0 = real ? x : 3e38f, x1 = real ? x : -3e38f, y0 = real ? y : 3e38f, y1 = real ? y : -3e38f;
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, o2, 64));
        x1 = fmaxf(x1, __shfl_xor(x1, o2, 64));
        y0 = fminf(y0, __shfl_xor(y0, o2, 64));
        y1 = fmaxf(y1, __shfl_xor(y1, o2, 64));
    }
    if (lane == 0) {
        const float4 c = box_circle(x0, x1, y0, y1, margin);
        if (pub) {   // write-through (csf_dev.h: part4): the NEXT pair launch, on the other stream, may start before this launch's end-of-kernel write-back
            unsigned long long *o = (unsigned long long *)&out[b];
            __hip_atomic_store(o, (unsigned long long)__float_as_uint(c.x) | ((unsigned long long)__float_as_uint(c.y) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o + 1, (unsigned long long)__float_as_uint(c.z) | ((unsigned long long)__float_as_uint(c.w) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            out[b] = c;
        }
    }
}


}  // namespace csf
