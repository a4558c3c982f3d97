// csf_engine.hip — host side of libcsf_hip.so: the C ABI of include/csf.h.
//
// Owns the SoA population in HBM, the HIP streams, the per-tick launch sequence and (for world > 1) the
// RCCL all-gather of the fp32 source records.  There is no CPU compute path in this library: every entry
// point that produces numbers launches the kernels of csf_pair.hip / csf_agent.hip.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <limits>
#include <string>
#include <unordered_map>
#include <vector>

#include "csf_dev.h"

using namespace csf;

namespace {

thread_local std::string g_create_error;

// RCCL is resolved at run time so that single-GPU use does not depend on it.
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) {
            err = std::string("cannot load librccl: ") + dlerror();
            return false;
        }
#define SYM(field, name)                                      \
    field = (decltype(field))dlsym(lib, name);                \
    if (!field) {                                             \
        err = std::string("librccl lacks symbol ") + name;    \
        return false;                                         \
    }
        SYM(GetUniqueId, "ncclGetUniqueId")
        SYM(CommInitRank, "ncclCommInitRank")
        SYM(CommDestroy, "ncclCommDestroy")
        SYM(AllGather, "ncclAllGather")
        SYM(AllReduce, "ncclAllReduce")
        SYM(Broadcast, "ncclBroadcast")
        SYM(GroupStart, "ncclGroupStart")
        SYM(GroupEnd, "ncclGroupEnd")
        SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
        return true;
    }
} g_rccl;

// CSF_DEBUG_POISON=1 (read at the first csf_create of the process): a debugging aid for loads whose safety is a property of the
// caller's allocation.  Every device buffer gets a red zone of POISON_GUARD bytes of 0xFF behind it (NaN as float or double, -1 as an
// integer: a load that runs past the end meets poison instead of a neighbour's data or a fault), the slots behind the population and
// the records no slot owns are poisoned instead of zeroed (upload_all, alloc_all) - a kernel whose RESULT depends on any of them
// shows NaN in the suite.  DESIGN.md section 3 lists the loads that are unconditional by design.
bool g_poison = false;
constexpr size_t POISON_GUARD = 1 << 16;

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        release();
        n = count;
        if (count == 0) return hipSuccess;
        const size_t guard = g_poison ? POISON_GUARD : 0;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T) + guard);
        // the clear runs on the NULL stream, which the engine's non-blocking streams do not wait for: finish it here
        if (e == hipSuccess) e = hipMemset(p, 0, count * sizeof(T));
        if (e == hipSuccess && guard) e = hipMemset((char *)p + count * sizeof(T), 0xFF, guard);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        return e;
    }
    // CSF_DEBUG_POISON: elements [from, n) become 0xFF bytes
    hipError_t poison_from(size_t from) {
        if (!g_poison || !p || from >= n) return hipSuccess;
        hipError_t e = hipMemset(p + from, 0xFF, (n - from) * sizeof(T));
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        return e;
    }
    // grow-only scratch: contents undefined
    hipError_t reserve(size_t count) { return count <= n ? hipSuccess : alloc(count); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

// Measurement and test knobs.  Read from the environment ONCE per engine, in csf_create: nothing on the tick path or in
// a re-binning calls getenv.
struct Knobs {
    double far_eps = 5.9604644775390625e-8;   // CSF_FAR_EPS: bound of the far-field / reach cull (2^-24; 0: off)
    bool reach = true;                        // CSF_REACH=0: no per-pair reach test
    std::string trace_blocks;                 // CSF_TRACE_BLOCKS=<file>: per-wave timeline of the last pair launch
    std::string trace_agent;                  // CSF_TRACE_AGENT=<file>: eight time stamps per wave of the last per-agent launch
    int fake_rank = 0, fake_world = 1;        // CSF_FAKE_SHARD=r/w: only rank r's receiver block of w (timing aid)
    int nsplit = 0;                           // CSF_NSPLIT: source chunks of the pair grid (0: the engine's choice)
    int dyn_recv = -1, rpb = 0;               // CSF_DYN_RECV, CSF_RPB
    int wide = -1;                            // CSF_WIDE: workgroups of 8 waves on tiles of 2048 sources (-1: the engine's choice)
    int pair_variant = -1;                    // CSF_PAIR_VARIANT (-1: by population size)
    int64_t rebin_ticks = 64;                 // CSF_REBIN_TICKS: ticks between two re-binnings (1 .. 120; the tests that step 40 - 48 ticks "across a re-binning" pin 32)
    int fused_mid = 1;                        // CSF_FUSED_MID=0: mid-size populations take a pair launch and a per-agent launch per tick (csf_mid.hip: one launch)
    int mid_below = 0;                        // CSF_MID_BELOW: ... for populations smaller than this (0: by vehicle class, mid_below_for)
    int mid_group = 0;                        // CSF_MID_GROUP: road users per workgroup of it, 4 / 8 / 16 / 32 (0: about one workgroup per CU)
    int chase = 1;                            // CSF_CHASE: 0 the per-agent launch behind the pair launch, 2 beside it (enqueue_chase_tick), 1 [default] whichever the engine measures faster on its first stretch of eligible ticks
    std::string chase_clock;                  // CSF_CHASE_CLOCK=<file>
    int chase_gate_pct = 85;                  // CSF_CHASE_GATE: per cent of the pair workgroups through before the per-agent kernel is let go
    int fused_small = 1;                      // CSF_FUSED_SMALL=0: a handful of road users take the general path too (csf_agent.hip: small_tick_kernel)
    int segments = -1;                        // CSF_SEGMENTS
    int seg_grid = 1;                         // CSF_SEG_GRID=0: one launch per parameter set (class-segmented order) instead of one grid
    int far_tight = 1;                        // CSF_FAR_TIGHT=0: the far-field radius prices all n sources at the cut (ln(n / eps) / kappa) also where candidate lists say how many a receiver can meet
    int clist = 1;                            // CSF_CLIST=0: receivers in binned order walk every tile of their chunk (no candidate lists)
    int recv_binned = -1;                     // CSF_RECV_BINNED
    int64_t rebin_churn = 4000;               // CSF_REBIN_CHURN
    int hole_reuse = 1;                       // CSF_HOLE_REUSE: an arrival takes the slot of a road user that left nearby (HoleIndex)
    double hole_dist = 1.0;                   // CSF_HOLE_DIST: ... within this many median batch radii of the batch's centre
    bool incremental = true;                  // CSF_INCREMENTAL=0: population changes through the host mirror
    int comm_second = -1;                     // CSF_COMM_STREAM=second / main (-1: the communicator times both on its first tick and keeps the faster)
    double fov_band = 1.0;                    // CSF_FOV_BAND: scale of the rounding band of the field-of-view test (0: every pair decided in fp32, as in round 3)
    double rnear = 1.0;                       // CSF_RNEAR: pairs closer than this (m) are evaluated from the precise records (0: none)
    int road_grid = -1;                       // CSF_ROAD_GRID: 1 the lattice of csf_road.hip for any road, 0 never (-1: large networks)
    double road_cell = 0.0;                   // CSF_ROAD_CELL: edge of its cells in m (0: 16)
    void read() {
        auto geti = [](const char *name, int dflt) {
            const char *v = getenv(name);
            return v ? atoi(v) : dflt;
        };
        if (const char *v = getenv("CSF_FAR_EPS")) far_eps = atof(v);
        reach = geti("CSF_REACH", 1) != 0;
        if (const char *v = getenv("CSF_TRACE_BLOCKS")) trace_blocks = v;
        if (const char *v = getenv("CSF_TRACE_AGENT")) trace_agent = v;
        if (const char *v = getenv("CSF_FAKE_SHARD")) {
            int fr = 0, fw = 1;
            if (sscanf(v, "%d/%d", &fr, &fw) == 2 && fw > 1 && fr >= 0 && fr < fw) fake_rank = fr, fake_world = fw;
        }
        // launch-shape knobs whose A/B is settled (every setting but the default measured slower): honoured only
        // with CSF_EXPERT=1, so that a stray variable cannot change the product's grid
        const bool expert = geti("CSF_EXPERT", 0) != 0;
        auto gete = [&](const char *name, int dflt) { return expert ? geti(name, dflt) : dflt; };
        nsplit = gete("CSF_NSPLIT", 0);
        dyn_recv = gete("CSF_DYN_RECV", -1);
        rpb = gete("CSF_RPB", 0);
        wide = gete("CSF_WIDE", -1);
        pair_variant = geti("CSF_PAIR_VARIANT", -1);
        chase = geti("CSF_CHASE", 1);
        if (const char *v = getenv("CSF_CHASE_CLOCK")) chase_clock = v;
        chase_gate_pct = std::max(0, std::min(100, geti("CSF_CHASE_GATE", 85)));
        fused_small = geti("CSF_FUSED_SMALL", 1);
        fused_mid = geti("CSF_FUSED_MID", 1);
        mid_below = geti("CSF_MID_BELOW", 0);
        mid_group = geti("CSF_MID_GROUP", 0);
        rebin_ticks = std::max(1, std::min(120, geti("CSF_REBIN_TICKS", 64)));
        segments = geti("CSF_SEGMENTS", -1);
        recv_binned = geti("CSF_RECV_BINNED", -1);
        clist = geti("CSF_CLIST", 1);
        far_tight = geti("CSF_FAR_TIGHT", 1);
        seg_grid = gete("CSF_SEG_GRID", 1);
        rebin_churn = std::max(1, geti("CSF_REBIN_CHURN", 4000));
        hole_reuse = geti("CSF_HOLE_REUSE", 1);
        if (const char *v = getenv("CSF_HOLE_DIST")) hole_dist = atof(v);
        incremental = geti("CSF_INCREMENTAL", 1) != 0;
        if (const char *v = getenv("CSF_COMM_STREAM")) comm_second = std::string(v) == "second" ? 1 : 0;
        if (const char *v = getenv("CSF_RNEAR")) rnear = atof(v);
        if (const char *v = getenv("CSF_FOV_BAND")) fov_band = std::max(0.0, atof(v));
        road_grid = geti("CSF_ROAD_GRID", -1);
        if (const char *v = getenv("CSF_ROAD_CELL")) road_cell = atof(v);
    }
};

}  // namespace

struct csf_engine {
    Dev d{};
    Knobs knobs;
    int device = 0;
    // parameter sets (csf_set_param_classes): classes[0] is d.p; h_cls[slot] the set of a road user
    std::vector<csf_params> classes;
    std::vector<double> class_kappa;     // far_kappa of every set (the grid search is done once per table)
    // Class-segmented order (several parameter sets, binned): every set is a run of places starting at a multiple of 64,
    // and the pair term is one launch of the culling kernel per run with that set's constants (rebin, launch_pair_all).
    struct Segment {
        int32_t cls;
        int64_t beg, end;                 // places
        int32_t chunk_units, n_split, part_base;
        PairConsts pc;
    };
    std::vector<Segment> segs;
    // ... the runs of the TwoD-field sets as ONE grid (csf_pair.hip: SEG): their table on the device, the grid's y extent
    DevBuf<SegDev> segtab;
    std::vector<SegDev> h_segtab;
    int seg_total_by = 0;
    int32_t sent_slot = 0;               // index of a record that is a sentinel for ever
    std::vector<uint8_t> h_cls;
    DevBuf<csf_params> ptab;
    DevBuf<PairConsts> pctab;
    DevBuf<double> pbtab;
    DevBuf<uint8_t> cls;
    bool classes_dirty = true;
    int64_t cap = 0;        // slots: the caller's capacity + head room for arrivals between two re-binnings (csf_create)
    int64_t cap_user = 0;   // road users the caller may have at once
    hipStream_t main = nullptr, comm = nullptr;
    // The members of a loopback group share ONE main stream (ticks and exchanges in order).  It belongs to all of them: the
    // handle lives as long as any member does, whichever is destroyed first (until round 5 it was the first member's, the
    // others were told when that one went - unless another member had gone before and cleared the lists through which they
    // would have been told: a dangling handle, which hipStreamSynchronize happened to tolerate and hipStreamQuery did not)
    struct StreamHold {
        hipStream_t s = nullptr;
        ~StreamHold() {
            if (s) (void)hipStreamDestroy(s);
        }
    };
    std::shared_ptr<StreamHold> main_hold;
    hipEvent_t ev_integ = nullptr, ev_gather = nullptr;
    std::string err;

    // host mirror (authoritative only while `host_ahead`; the device is authoritative after a tick)
    std::vector<double> h_s, h_vdes, h_znp, h_hx, h_hy, h_lti, h_ppsi, h_F;
    std::vector<int32_t> h_ptr, h_ti, h_dgood;
    std::vector<uint8_t> h_znav, h_zrid;
    std::vector<uint32_t> h_status;
    std::vector<std::vector<double>> h_q;  // per slot: rows of (x, y, stop) - the host is authoritative for the rows
    std::vector<std::vector<double>> h_script;   // per slot: rows (x, y, psi, v) of an UncontrolledVehicle's prescribed trajectory
    DevBuf<double> script;
    DevBuf<int64_t> sbeg;
    DevBuf<int32_t> slen;
    // The population: road user i of the caller's order lives in slot order[i].  Slots are what every device array is
    // indexed by.  csf_remove_agents on a live device copy only kills slots (they keep a sentinel record), csf_add_agents
    // reuses them; a full upload compacts the slots back into population order.
    std::vector<int32_t> order;
    // Dead slots.  `free_tail`: dead at the last re-binning, so their place in the binned order is in the tail of
    // sentinels behind the real batches - a road user spawned into one of them (or into a fresh slot below n_pad) joins a
    // tail batch of other newcomers and stretches no real batch's circle.  `free_recent`: retired since; their places are
    // inside real batches, so they are handed out only when nothing else is left, and move to `free_tail` at the next
    // re-binning.  `slack`: sentinel slots kept behind the population for arrivals (twice what the last period saw).
    std::vector<int32_t> free_tail, free_recent;   // free_tail: descending, so that pop_back hands out ascending slots
    int64_t live_at_rebin = 0, tail_used = 0;       // road users at the last re-binning; sentinel places handed out since
    int64_t tail_flushed = 0;                       // ... of which the device has seen this many (flush_pending)
    bool pend_inplace = false;                      // an arrival of the pending batch took a slot inside a real batch
    // HOLES (road users arriving and leaving every tick, SUMO co-simulation: intersection.py:458-634).  A road user that leaves
    // leaves a hole in a real batch of the binned order; an arrival that starts within that batch's circle can have the hole -
    // the circle does not grow, the sentinel tail does not fill up, and the order lasts its 64 ticks instead of three.  What the
    // host needs for that it reads back once per re-binning, without waiting: the place of every slot and the circles (85 KB at
    // N = 16 384).  The holes are kept in a lattice over the scene by the centre of their batch.
    struct HoleIndex {
        int32_t *pos = nullptr;                     // pinned: slot -> place at the last re-binning
        float4 *bnd = nullptr;                      // pinned: circle of every batch then (scene coordinates)
        std::vector<int32_t> hpos;                  // ... copied to ordinary memory when they have landed (the lookups
        std::vector<float4> hbnd;                   //     per arrival and departure are the host's hot loop under traffic)
        size_t pos_n = 0, bnd_n = 0;
        hipEvent_t ev = nullptr;
        bool pending = false, ready = false;        // the read-back is under way / the lattice is built
        int64_t places = 0;                         // places that held road users at the re-binning
        double x0 = 0, y0 = 0, cell = 1, inv_cell = 1, reach2 = 0, ox = 0, oy = 0;
        int nx = 0, ny = 0;
        struct Ent { int32_t slot; float x, y; };   // a hole: the slot and the centre of its batch (scene coordinates)
        static constexpr int CELL_CAP = 5;          // holes a lattice cell can list (one more stays in free_recent, unlisted)
        struct alignas(64) Cell { int32_t n; Ent e[CELL_CAP]; };   // one cache line: the host looks a cell up per arrival
        std::vector<Cell> cell_tab;                 // [ny][nx]
        int64_t taken = 0;                          // arrivals that took a leaver's slot (csf_holes_taken: tests, tools)
    } holes;
    std::vector<int32_t> add_slots;                 // csf_add_agents: the slots of the call's arrivals
    int64_t pend_tail_spawns = 0;                   // arrivals of the pending batch that went to the sentinel tail
    DevBuf<unsigned> ticket;                        // patch_kernel: which workgroup finishes last
    // grow-only device scratch of the single-piece entry points (csf_untracked, csf_update_*, csf_count_pairs): no
    // hipMalloc / hipFree per call
    DevBuf<uint8_t> scratch_u8;
    DevBuf<int32_t> scratch_i32;
    DevBuf<double> scratch_f64;
    DevBuf<unsigned long long> scratch_cnt;
    bool tail_tracked = false;                      // the places of the sentinel tail are known (binned single engine)
    std::vector<uint8_t> h_alive;
    bool order_dirty = true;               // the device copy of `order` is stale
    bool incremental = true;               // csf_set_incremental
    int64_t q_top = 0;                     // rows of the queue slab in use
    int64_t churn = 0;                     // road users spawned into free slots since the last re-binning
    std::vector<double> h_road;            // per vertex (x, y, F0, sigma)
    // the lattice over a large road network (csf_road.hip; build_road_grid): what was built, and for which vertex set
    uint64_t road_version = 1, rg_version = 0;
    double rg_cell = 0, rg_gx0 = 0, rg_gy0 = 0, rg_ox = 0, rg_oy = 0;
    int rg_nx = 0, rg_ny = 0;
    DevBuf<float4> rg_v;
    DevBuf<int32_t> rg_start;
    DevBuf<float> rg_c;
    bool dirty = true;                     // host mirror changed since the last upload
    bool device_ahead = false;             // ticks ran since the last download

    DevBuf<double> s, vdes, q, znp, hx, hy, lti, ppsi, F, hist;
    DevBuf<int64_t> qbeg;
    DevBuf<int32_t> ptr, ti, dgood, qlen, order_dev;
    DevBuf<uint8_t> znav, zrid, alive;
    // Population changes on a current device copy are collected here and applied by ONE kernel launch at the next
    // device call (flush_pending).  The kernel reads its records from pinned, device-visible host memory: a ring of four
    // buffers, each guarded by an event (the kernel that read a buffer has finished long before the ring comes round).
    struct Pending {
        std::vector<int32_t> retire;
        std::vector<SpawnRec> spawn;       // qbeg relative to `rows` until the flush
        std::vector<QueueRec> requeue;     // the same
        std::vector<double> rows;          // (x, y, stop) rows of new and replaced queues
        bool empty() const { return retire.empty() && spawn.empty() && requeue.empty(); }
    } pend;
    // [cap] per slot, ONE 16-byte entry (a leaver or an arrival touches all four of its slot: four arrays were four cache misses each,
    // 819 times per tick at 5 % churn): index of the slot in the pending lists above and in free_recent, -1: not there
    struct SlotIdx {
        int32_t spawn = -1, requeue = -1, retire = -1, recent = -1;
    };
    std::vector<SlotIdx> sidx;
    std::vector<uint8_t> dev_alive;        // [cap] what d.alive holds on the device (as of the last flush or upload)
    struct PinnedSlot {
        void *host = nullptr, *dev = nullptr;
        size_t bytes = 0;
        hipEvent_t done = nullptr;
        bool busy = false;
    };
    PinnedSlot pinned[4];
    int pinned_next = 0;
    DevBuf<uint32_t> status;
    DevBuf<float4> rec, recs, recg, recb, rv, kat4, bnd, bnd2;
    // the fused tick of mid-size populations (csf_mid.hip): the other halves of the record double buffers, the two snapshots of
    // (x, y, psi), the arrival counters; mid_synced: the other halves hold this tick's sentinels and the current snapshot holds
    // this tick's state (any other path that writes records or state clears it)
    DevBuf<float4> rec_alt, recg_alt;
    DevBuf<float2> rec2_alt;
    DevBuf<double> src64_a, src64_b;
    bool mid_synced = false, mid_cur_is_a = true;
    int64_t mid_ticks = 0;
    // the per-agent launch beside the pair launch (enqueue_chase_tick): the binned copy's other half, the arrival counters
    // ([groups of 64 slots]) and the gate's / the error word, whether the last tick took this path (the two streams then have to
    // meet before anything else runs), ticks since the counters were cleared, which stream the next pair launch goes to
    DevBuf<float4> recs_alt;
    DevBuf<unsigned> chase_cnt, chase_misc;
    DevBuf<unsigned long long> chase_clock;   // CSF_CHASE_CLOCK=<file>: stamps of the last 64 side-by-side ticks, written at csf_destroy
    bool dirty_layout_for_warm() const { return dirty || !segs.empty() || classes.size() != 1; }   // (the warm pair launch of chase_alloc takes the plain single-set launch)
    // CSF_TIME_POP=1 (measurement aid): nanoseconds inside the population calls, by part, printed at csf_destroy
    bool time_pop = false;
    int64_t tp_ns[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp_calls = 0;
    std::vector<int32_t> remove_sorted;   // csf_remove_agents: the listed indices sorted, when the caller's are not
    bool chase_prev = false, chase_resume = false;   // resume: the counters and halves are those of the last side-by-side tick (mid_synced says nothing else wrote records since)
    uint32_t chase_round = 0;
    int chase_parity = 0;
    int64_t chase_ticks = 0, chase_checked = 0;
    // CSF_CHASE=1: decided by measurement, once per engine (chase_take): three periods between re-binnings - in turn, side by side, in turn -, bracketed by events on the
    // main stream - whether the two streams got hardware queues of their own is the runtime's business, and without them the
    // side-by-side tick is the slower one
    int chase_state = 0;                // 0 undecided, 1 side by side, -1 in turn
    int cal_phase = 0;                  // 0 idle, 1 .. 3 the periods (in turn, side by side, in turn), 4 waiting for the last event
    hipEvent_t cal_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    double chase_cal_us[2] = {0.0, 0.0};
    int64_t cal_ticks[3] = {0, 0, 0};   // ticks of every period of the measurement
    DevBuf<float2> borg;
    DevBuf<float2> rvo;          // origins of the road-vertex tiles
    DevBuf<int32_t> pos;
    bool bounds_fresh = false;   // d.bnd describes the current records (else: recompute before the pair kernel)
    DevBuf<int32_t> perm, sort_vals, rlist;
    DevBuf<uint32_t> sort_keys, sort_keys_out;
    DevBuf<uint8_t> sort_tmp;
    int64_t ticks_since_rebin = 0;
    // integration steps since the last re-binning that no pair launch went with (csf_apply_forces, csf_replay_forces on forces
    // from elsewhere): the candidate tile lists and the stretch of the circles allow for rebin_ticks + 2 steps of motion, so
    // these count towards the next re-binning like ticks do
    int64_t moved_unbinned = 0;
    bool pair_since_move = false;
    DevBuf<float2> rec2, recs2, part, froad, kat2;
    // candidate tiles of every receiver group (large populations, csf_dev.h: Dev::clist): rebuilt at every re-binning
    DevBuf<float4> tcirc;
    DevBuf<uint16_t> clist;
    DevBuf<int32_t> ccount;
    DevBuf<unsigned> far_stat;   // [2] what launch_candidate_lists found for the far-field bound (tighten_far_bound)
    double far_met = 0.0, far_tail = 0.0, far_T = 0.0;   // ... and what became of it: sources met, tail (x f_0), T in use (0: the plain bound)
    DevBuf<float2> rorg;         // the origin every precise record is relative to, by slot
    DevBuf<float2> reclo;        // what the record's position left over in fp32 (csf_dev.h); travels with the records
    DevBuf<float4> xbuf;         // the exchange records of a sharded run (csf_dev.h: xbuf), allocated with the shard layout
    bool xbuf_fresh = false;     // other ranks' exchange records have arrived and are not yet spread to rec / reclo / rec2
    bool state_all_current = true;   // every live slot's fp64 state on this device is current (false once a rank has ticked)
    // Field-of-view decisions within fp32 rounding of an edge (csf_dev.h: PairConsts::fovA, EdgeRec; set_fov_band): the
    // largest |coordinate| relative to the scene origin at the last upload (road users, prescribed trajectories, arrivals
    // since), and the integration steps since - nobody moves farther than t_s * v_max per step
    double coord_bound0 = 0.0;
    double *bound_pin = nullptr;     // pinned: where the positions land when that bound is measured again (set_fov_band)
    int64_t moves = 0;
    int64_t small_ticks = 0;         // ticks run by the one-wave kernel (csf_small_ticks)
    int64_t slab_rewrites = 0;       // compact_slab calls (upload_queues sizes the slab by them)
    bool bound_stale = false;        // positions moved without a speed clamp (csf_replay_forces with fix_speed): measure them
    uint32_t edge_stamp = 0;
    DevBuf<EdgeRec> edge;
    DevBuf<unsigned> edge_n;         // [0] ring counter, [1] near / marginal pairs that could not be noted (csf_near_dropped)
    DevBuf<int32_t> edge_head;
    double far_kappa = 0.0;   // lower bound of the field's decay rate (far_kappa())
    double last_gather_ms = 0.0;  // all-gather time accumulated by the last csf_profile_read
    bool ev_gather_recorded = false;
    void *snap_host = nullptr;  // csf_get_tick: pinned, device-mapped staging buffer
    double *snap_dev = nullptr;
    size_t snap_bytes = 0;
    DevBuf<uint64_t> trace;   // CSF_TRACE_BLOCKS (measurement aid)
    DevBuf<uint64_t> atrace;  // CSF_TRACE_AGENT (measurement aid)
    size_t trace_words = 0;

    // sharding
    int rank = 0, world = 1;
    ncclComm_t nccl = nullptr;
    bool gather_pending = false;
    // where the all-gather of a tick is issued: in stream order on the main stream, or on the second stream beside the
    // destination-force phase of the next tick.  Chosen by the communicator itself on its first tick
    // (calibrate_comm_stream: both orders timed on the REAL communicator, the same answer on every rank) unless
    // CSF_COMM_STREAM says which.
    bool comm_second = false, comm_calibrated = false;
    float comm_cal_us[2] = {0.f, 0.f};       // what the calibration measured per tick: stream order, second stream

    // loopback rehearsal of the sharded path: `world` engines of one process on one device share a stream and exchange
    // their record blocks with device-to-device copies where the ranks of a real run call ncclAllGather
    std::vector<csf_engine *> group;
    bool loopback = false;

    // profiling: a fixed pool of event slots, recycled in order (the oldest slot is resolved into the running sums
    // before it is reused, so stepping with profiling left on holds a bounded number of events)
    int profile = 0;             // 0 off, k > 0: time the kernels of every k-th tick
    struct ProfSlot {
        hipEvent_t ev[8] = {};   // pair, road, agent: start / end of the kernel itself; all-gather: recorded around it
        bool pair = false, road = false, agent = false, gather = false;
    };
    std::vector<ProfSlot> prof_pool;
    size_t prof_issued = 0, prof_resolved = 0;
    double prof_ms[4] = {0, 0, 0, 0};   // pair, road, agent, gather
    int64_t prof_cnt[4] = {0, 0, 0, 0}; // launches behind each sum
    int64_t prof_ticks = 0;             // sampled ticks issued (the kernels beside the pair kernel are timed on every 8th)
    std::vector<float> prof_us[4];      // per sampled launch and kernel - pair, road, per-agent, all-gather - (at most PROF_KEEP of each)
};

// (defined with the side-by-side tick, used by upload_all)
static bool chase_shape(const csf_engine *e);
static int chase_alloc(csf_engine *e);

namespace {

int fail(csf_engine *e, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (e) e->err = buf;
    else g_create_error = buf;
    return code;
}

#define HIPCHK(e, call)                                                                           \
    do {                                                                                          \
        hipError_t _r = (call);                                                                   \
        if (_r != hipSuccess)                                                                     \
            return fail(e, CSF_E_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

#define NCCLCHK(e, call)                                                                          \
    do {                                                                                          \
        ncclResult_t _r = (call);                                                                 \
        if (_r != ncclSuccess)                                                                    \
            return fail(e, CSF_E_COMM, "%s failed: %s", #call, g_rccl.GetErrorString(_r));        \
    } while (0)

const int NS_OF[7] = {5, 5, 6, 4, 5, 4, 8};
// Ticks between two re-binnings.  The circles of the batches are renewed EVERY tick from the positions as they are (the pair
// kernel emits them), so a stale order costs only what the batches' members drift apart: the pair kernel takes the same
// 105 us with 32, 48 and 64 ticks and 0.3 - 0.5 us more with 96 / 128, while the re-binning itself (~40 us of sort, rebase
// and launch gaps) is paid half as often: 64 (round 4: +0.5 % over 32; profiles/r4_rebin_period_ab.txt).
constexpr int64_t REBIN_TICKS = 64;     // (default of Knobs::rebin_ticks)
int32_t pair_variant_for(const csf_engine *e, int64_t n);   // (with rebin, below)

double limit_angle_h(double th) {  // utils.py:124-139 (host: Vehicle.__init__, vehicle.py:154-155)
    const double PI = 3.141592653589793238462643383279502884;
    th = std::floor(th / (2 * PI)) * (-2 * PI) + th;
    if (th > PI) th -= 2 * PI;
    else if (th < -PI) th += 2 * PI;
    return th;
}

int check_params(csf_engine *e, const csf_params *p) {
    if (!p) return fail(e, CSF_E_ARG, "params is NULL");
    if (p->model < 0 || p->model > 6) return fail(e, CSF_E_ARG, "unknown model %d", p->model);
    if (p->model == CSF_PLANARBIKE) {
        const double sum_im = p->pb_poles[1] + p->pb_poles[3], prod_im = p->pb_poles[0] * p->pb_poles[3] + p->pb_poles[1] * p->pb_poles[2];
        if (std::fabs(sum_im) > 1e-12 || std::fabs(prod_im) > 1e-12 || !(p->pb_poles[0] < 0) || !(p->pb_poles[2] < 0))
            return fail(e, CSF_E_ARG, "PlanarBicycle needs two stable poles, real or a conjugate pair");
        if (!(p->l > 0)) return fail(e, CSF_E_ARG, "PlanarBicycle needs a wheelbase l > 0");
    }
    if (!(p->t_s > 0)) return fail(e, CSF_E_ARG, "t_s must be > 0");
    if (p->traj_len < 2) return fail(e, CSF_E_ARG, "traj_len must be >= 2 (int(30/t_s) in the reference)");
    if (p->priority_rule < 0 || p->priority_rule > 1) return fail(e, CSF_E_ARG, "unknown priority rule");
    if (!(p->hfov >= 0)) return fail(e, CSF_E_ARG, "hfov must be >= 0");
    return CSF_OK;
}

// Far-field radius of the TwoD field (vehicle.py:1560-1648).  |F| = f_0 exp(-rho q / sigma) with
// q = sqrt(1 - e^2 cos^2 phi), sigma = sga - sgb |sin(phi/2)|, all functions of s2 = sin^2(psi0 - psi) in [0, 1] and of
// phi: kappa = min q / sigma bounds the decay from below, so every source beyond
//     R = ln(n / far_eps) / kappa
// adds less than far_eps f_0 / n, and all of them together less than far_eps f_0, to a receiver's column sum.
// far_eps defaults to 2^-24 (half an fp32 ulp of f_0; CSF_FAR_EPS overrides, 0 switches the cull off): what is
// left out is below the rounding of the fp32 sum it would have been added to.  Batches of binned records
// whose bounding circle lies entirely beyond R are skipped by the pair kernel.
double far_kappa(const csf_params &p) {
    if (p.model == CSF_BICYCLE) return 0.0;
    double kappa = INFINITY;
    for (int i = 0; i <= 256; i++) {
        const double s2 = i / 256.0;
        const double e = p.e_0 - p.e_1 * s2, sga = p.sigma_0 + p.sigma_1 * s2, sgb = p.sigma_2 + p.sigma_3 * s2;
        for (int j = 0; j <= 512; j++) {
            const double phi = 3.141592653589793 * j / 512.0;
            const double q2 = 1.0 - e * e * std::cos(phi) * std::cos(phi), sigma = sga - sgb * std::sin(0.5 * phi);
            if (!(sigma > 0) || !(q2 > 0)) return 0.0;   // degenerate parameters: no usable bound
            kappa = std::min(kappa, std::sqrt(q2) / sigma);
        }
    }
    return std::isfinite(kappa) ? 0.98 * kappa : 0.0;  // the grid is fine but finite
}

double far_radius(double kappa, int64_t n, double far_eps) {
    if (!(far_eps > 0) || !(kappa > 0) || n < 1) return INFINITY;
    return std::log((double)n / far_eps) / kappa;
}

// Per-pair form of the same bound.  A pair adds f_0 exp(-x), x = rho q / sigma (vehicle.py:1628), and may be left out
// when x > T = ln(n / far_eps): then it is below far_eps f_0 / n like every source beyond the far-field radius.  With
// X = rho cos(phi) (the receiver along the source's heading), rho^2 q^2 = rho^2 - e^2 X^2 exactly, and
// sigma = sigma_a - sigma_b |sin(phi/2)| <= sigma_a - sigma_b (1 - cos phi) / 2 because |sin(phi/2)| <= 1; so
//     rho^2 - e^2 X^2 > T^2 (sigma_a - sigma_b / 2 + (sigma_b / 2) X / rho)^2   implies   x > T.
// The kernel evaluates this with one rsq per pair (csf_pair.hip: keep_x2); T carries a 0.2 % margin against the fp32
// rounding of the two sides.  At N = 16 384 in 200 m the test removes three of four pairs that pass the field of view.
// far-field radius and reach-test constants of one parameter set in a population of n road users
void set_far_consts(const Knobs &kn, const csf_params &p, double kappa, int64_t n, PairConsts &k) {
    const double eps = kn.far_eps;
    k.rfar = (float)far_radius(kappa, n, eps);
    // (sigma_b >= 0 is what the bound sigma <= sigma_a - sigma_b (1 - cos phi) / 2 rests on; the reference's setters enforce
    // it, the C ABI does not: a set with a negative sigma_2 / sigma_3 keeps the radius-only cull)
    const bool on = std::isfinite(k.rfar) && p.model != CSF_BICYCLE && n >= 1 && p.sigma_2 < p.sigma_0 && p.sigma_3 < p.sigma_1 &&
                    p.sigma_2 >= 0 && p.sigma_3 >= 0 && p.sigma_2 + p.sigma_3 >= 0 && kn.reach;
    k.reach = on;
    if (on) {
        const double T = std::log((double)n / eps) * 1.002;
        k.tA0 = (float)(T * (p.sigma_0 - 0.5 * p.sigma_2));
        k.tA1 = (float)(T * (p.sigma_1 - 0.5 * p.sigma_3));
        k.tB0 = (float)(T * 0.5 * p.sigma_2);
        k.tB1 = (float)(T * 0.5 * p.sigma_3);
    } else {
        k.tA0 = k.tA1 = k.tB0 = k.tB1 = 0.f;
    }
}

// set_far_consts with ln(n / eps) replaced by T
static void set_far_consts_T(const csf_params &p, double kappa, double T, PairConsts &k) {
    k.rfar = (float)(T / kappa);
    if (k.reach) {
        const double Tm = T * 1.002;
        k.tA0 = (float)(Tm * (p.sigma_0 - 0.5 * p.sigma_2));
        k.tA1 = (float)(Tm * (p.sigma_1 - 0.5 * p.sigma_3));
        k.tB0 = (float)(Tm * 0.5 * p.sigma_2);
        k.tB1 = (float)(Tm * 0.5 * p.sigma_3);
    }
}

void update_far_radius(csf_engine *e) {   // depends on the parameters and on the population size
    e->far_T = 0.0;
    PairConsts &k = e->d.pc;
    if (e->classes.size() > 1) {          // several parameter sets: the plain kernel, or one launch per set with its own
        k.rfar = INFINITY;                // constants (Segment::pc)
        k.reach = 0;
        k.tA0 = k.tA1 = k.tB0 = k.tB1 = 0.f;
        return;
    }
    set_far_consts(e->knobs, e->d.p, e->far_kappa, e->d.n, k);
}

// The rounding band of the fp32 field-of-view test (csf_dev.h: PairConsts::fovA ...; csf_field.h: keep_x2, tracked_m), for the
// pair launches that follow.  With u = 2^-24: a position in a tile is off by eps_p = u * (largest |coordinate|) per
// component (offset + origin rounded once), a heading (cos, sin) by u per component; t = rho cos(bearing) is then off by
// < 3 eps_p + 5 u rho and g = t|t| + chs rho^2 by < 12 eps_p rho + 18 u rho^2.  Twice that, with rho <= r2 / 16 + 4:
//     band = (36 u + 1.5 eps_p) r2 + 96 eps_p;   side test (priority to the right): 0.625 u r2 + 6 eps_p + 40 u
// (tracked_m: the plain kernels).  The packed test of the cull-first kernel (keep_x2) compares cos(bearing) = t rsq(rho^2)
// with cos(hfov / 2): off by < 3 eps_p / rho + 8 u (the rsq included); twice that is its band, and that of sin(bearing).
// A pair formed from the precise records (offsets of at most a few metres from origins whose difference is exact) has
// eps_p = u * (largest offset), and its band is evaluated with rho itself: 24 eps_p rho + 36 u r2; 6 eps_p + 10 u rho.
int set_fov_band(csf_engine *e) {
    Dev &d = e->d;
    const double u = 5.9604644775390625e-8;
    if (e->bound_stale) {   // (single device: every slot's state is here)
        // (through a pinned buffer: the first copy into pageable memory of a process costs ~9 ms - the runtime sets its staging
        // path up - and a population that is only ever stepped met it here, in the middle of a run: tick 4 096 of config 2)
        if (!e->bound_pin) HIPCHK(e, hipHostMalloc((void **)&e->bound_pin, 2 * (size_t)e->cap * sizeof(double), hipHostMallocDefault));   // (csf_create made it)
        HIPCHK(e, hipMemcpyAsync(e->bound_pin, e->s.p, 2 * (size_t)e->cap * sizeof(double), hipMemcpyDeviceToHost, e->main));
        HIPCHK(e, hipStreamSynchronize(e->main));
        const double *xy = e->bound_pin;
        double cb = e->coord_bound0;
        for (int32_t a : e->order) {
            const double bx = std::fabs(xy[(size_t)a] - d.ox), by = std::fabs(xy[(size_t)e->cap + (size_t)a] - d.oy);
            if (std::isfinite(bx) && std::isfinite(by)) cb = std::max({cb, bx, by});
        }
        e->coord_bound0 = cb;
        e->moves = 0;
        e->bound_stale = false;
    }
    double vmax = 0;
    for (const csf_params &c : e->classes)
        vmax = std::max({vmax, std::fabs(c.v_max_riding[0]), std::fabs(c.v_max_riding[1]), std::fabs(c.v_max_walk)});
    const double step = d.p.t_s * vmax * 1.01 + 1e-4;
    // receivers in binned order work relative to the origin of their group: up to twice the bound from the scene origin
    double cmax = e->coord_bound0 + step * (double)(e->moves + 2) + 1.0;
    // receivers in binned order work relative to the origin of their group: at most twice the bound from the scene origin
    // (a tighter one - far-field radius + the extent of a group and of a batch - would need the circles, which only the
    // device knows; the band only decides how many pairs take the exact path, a few per thousand receivers either way)
    if (d.recv_binned) cmax *= 2.0;
    const double eps_p = u * cmax * 1.01 + 4 * u;
    // (full circle: chs = 4, g > 3 r2 - never inside a band that small)
    PairConsts &k = d.pc;
    const double sc = e->knobs.fov_band;
    k.fovA = (float)(sc * (36 * u + 1.5 * eps_p));
    k.fovB = (float)(sc * 96 * eps_p);
    k.sideA = (float)(sc * 0.625 * u);
    k.sideB = (float)(sc * (6 * eps_p + 40 * u));
    k.fovT0 = (float)(sc * 16 * u);
    k.fovT1 = (float)(sc * 6 * eps_p);
    // whole batches are classified against the field-of-view cone with a margin of 1e-4 in the cosine (csf_pair.hip:
    // classify_batch): the bearing of a source nearer than 2.9 u (coordinates) / 5e-5 is not known that well in fp32, so a
    // batch whose circle comes closer goes to the per-lane test
    k.clsk = (float)(2.9 * u * 1.01 / 5e-5);
    // offsets: a quarter-metre grid of origins + what a road user covers between two re-binnings (REBIN_TICKS steps + a few;
    // arrivals take their position as their origin)
    const double off = 0.25 + step * (double)(e->knobs.rebin_ticks + 8);
    const double eps_o = u * off * 1.5;
    k.fovP1 = (float)(sc * 24 * eps_o);
    k.fovP2 = (float)(sc * 36 * u);
    k.sideP0 = (float)(sc * 6 * eps_o);
    k.sideP1 = (float)(sc * 10 * u);
    for (csf_engine::Segment &sg : e->segs) {
        sg.pc.fovA = k.fovA, sg.pc.fovB = k.fovB, sg.pc.sideA = k.sideA, sg.pc.sideB = k.sideB;
        sg.pc.fovT0 = k.fovT0, sg.pc.fovT1 = k.fovT1, sg.pc.clsk = k.clsk;
        sg.pc.fovP1 = k.fovP1, sg.pc.fovP2 = k.fovP2, sg.pc.sideP0 = k.sideP0, sg.pc.sideP1 = k.sideP1;
    }
    d.state_current = e->state_all_current ? 1 : 0;
    d.edge = e->edge.p;
    d.edge_stamp = ++e->edge_stamp;
    d.keep_lo = (e->world > 1 || e->loopback || e->nccl != nullptr) ? 1 : 0;
    return CSF_OK;
}

// exp(A) of a small dense matrix (n <= 4, row major): scaling and squaring of a degree-18 Taylor polynomial
void expm_small(int n, const double *A, double *E) {
    double nrm = 0;
    for (int i = 0; i < n * n; i++) nrm = std::max(nrm, std::fabs(A[i]));
    int sq = 0;
    while (nrm * n > 0.25 && sq < 60) nrm *= 0.5, sq++;
    const double sc = std::ldexp(1.0, -sq);
    double M[16], T[16], P[16];
    for (int i = 0; i < n * n; i++) M[i] = A[i] * sc;
    for (int i = 0; i < n * n; i++) E[i] = (i / n == i % n) ? 1.0 : 0.0, P[i] = E[i];
    for (int k = 1; k <= 18; k++) {
        for (int r = 0; r < n; r++)
            for (int c = 0; c < n; c++) {
                double acc = 0;
                for (int q = 0; q < n; q++) acc += P[r * n + q] * M[q * n + c];
                T[r * n + c] = acc / k;
            }
        for (int i = 0; i < n * n; i++) P[i] = T[i], E[i] += T[i];
    }
    for (int it = 0; it < sq; it++) {
        for (int r = 0; r < n; r++)
            for (int c = 0; c < n; c++) {
                double acc = 0;
                for (int q = 0; q < n; q++) acc += E[r * n + q] * E[q * n + c];
                T[r * n + c] = acc;
            }
        for (int i = 0; i < n * n; i++) E[i] = T[i];
    }
}

// PlanarBicycle (dynamics.py:178-258, 1167-1226).  x = (delta, psi), A(v) = [[0, 0], [a, 0]] with a = v / l, B = (1, 0)^T,
// K_x = (k1, k2) places the class's poles p1, p2: s^2 + k1 s + k2 a = (s - p1)(s - p2), so k1 = -(p1 + p2), k2 = p1 p2 / a.
// With z = (a delta, psi) the closed loop reads z' = M z + (a, 0)^T u, M = [[-k1, -p1 p2], [1, 0]] - no speed in M.
// The reference sets K_u = 1 / psi_sim(T), psi_sim the yaw at the end of a simulated response (T = 0 .. 9.99 in steps of
// 0.01, input 0 for ten samples then 1, first-order hold) of the loop with K_u = 1: psi_sim = a G with G the same response
// of z' = M z + (1, 0)^T u.  Hence a K_u = 1 / G and the controlled loop is z' = M z + (1 / G, 0)^T psi_d at EVERY speed:
// one exact step z+ = E z + Gamma psi_d is derived here once; the kernel scales delta by a on the way in and out.
void derive_planarbike(const csf_params &p, double pb[7]) {
    const double k1 = -(p.pb_poles[0] + p.pb_poles[2]), w2 = p.pb_poles[0] * p.pb_poles[2] - p.pb_poles[1] * p.pb_poles[3];
    auto step_matrices = [&](double dt, double Ad[4], double Bd0[2], double Bd1[2]) {   // first-order hold, input (1, 0)^T
        double Mx[16] = {0}, Ex[16];
        Mx[0] = -k1 * dt; Mx[1] = -w2 * dt; Mx[2] = dt;
        Mx[4] = dt;
        Mx[11] = 1.0;
        expm_small(4, Mx, Ex);
        Ad[0] = Ex[0]; Ad[1] = Ex[1]; Ad[2] = Ex[4]; Ad[3] = Ex[5];
        Bd1[0] = Ex[3]; Bd1[1] = Ex[7];
        Bd0[0] = Ex[2] - Bd1[0]; Bd0[1] = Ex[6] - Bd1[1];
    };
    double Ad[4], Bd0[2], Bd1[2];
    step_matrices(0.01, Ad, Bd0, Bd1);                         // from_pole_placement's own t_s = 0.01, t_end = 10
    double z0 = 0, z1 = 0;
    for (int i = 1; i < 1000; i++) {
        const double u0 = (i - 1) >= 10 ? 1.0 : 0.0, u1 = i >= 10 ? 1.0 : 0.0;
        const double n0 = Ad[0] * z0 + Ad[1] * z1 + Bd0[0] * u0 + Bd1[0] * u1;
        const double n1 = Ad[2] * z0 + Ad[3] * z1 + Bd0[1] * u0 + Bd1[1] * u1;
        z0 = n0, z1 = n1;
    }
    const double G = z1;
    step_matrices(p.t_s, Ad, Bd0, Bd1);                        // forced_response over [0, t_s], input psi_d at both ends
    for (int i = 0; i < 4; i++) pb[i] = Ad[i];
    pb[4] = (Bd0[0] + Bd1[0]) / G;
    pb[5] = (Bd0[1] + Bd1[1]) / G;
    pb[6] = std::exp(-p.k_p_v * p.t_s);                        // dynamics.py:156
}

// what the pair kernels need of one parameter set
void derive_pair_consts(const csf_params &p, PairConsts &k, double rnear) {
    k.sg0 = (float)p.sigma_0;
    k.sg1 = (float)p.sigma_1;
    k.sg2 = (float)p.sigma_2;
    k.sg3 = (float)p.sigma_3;
    k.e0 = (float)p.e_0;
    k.e1 = (float)p.e_1;
    k.kexp = 1.4426950408889634f;
    const double PI_ = 3.141592653589793;
    double ch = std::cos(0.5 * std::min(p.hfov, 2 * PI_));
    k.chs = (float)(p.hfov <= PI_ ? -ch * ch : ch * ch);
    if (p.hfov >= 2 * PI_) k.chs = 4.0f;  // full circle: every bearing is inside (t|t| + 4 rho^2 > 0)
    k.ch = (float)ch;
    k.chm = k.ch - 1e-4f;
    k.chp = k.ch + 1e-4f;
    k.chk = p.hfov >= 2 * PI_ ? -2.0f : (float)ch;      // (keep_x2: cos(bearing) > chk)
    k.p2r = p.priority_rule == CSF_P2R;
    if (p.model == CSF_BICYCLE) {
        k.lf0 = (float)std::log2(p.p_0 / p.p_decay);
        k.ipd = (float)(1.0 / p.p_decay);
        k.f0_zero = 0;
    } else {
        k.lf0 = p.f_0 > 0 ? (float)std::log2(p.f_0) : -INFINITY;
        k.ipd = 0.f;
        k.f0_zero = p.f_0 == 0.0;
    }
    k.full_circle = p.hfov >= 2 * PI_;
    k.fov_classify = 1;   // any field of view: csf_pair.hip guards the wide ones (set to 0 to fall back to exact tests)
    k.rfar = INFINITY;
    k.tA0 = k.tA1 = k.tB0 = k.tB1 = 0.f;
    k.reach = 0;
    k.rnear = (float)std::max(0.0, rnear);
    k.rnear2 = k.rnear * k.rnear;
    k.rn2big = 1e20f * k.rnear2;
}

void derive_consts(csf_engine *e) {
    const csf_params &p = e->d.p;
    if (p.model == CSF_PLANARBIKE) derive_planarbike(p, e->d.pb);
    derive_pair_consts(p, e->d.pc, e->knobs.rnear);
    e->far_kappa = far_kappa(p);
    update_far_radius(e);
    if (e->classes.empty()) e->classes.push_back(p);
    e->classes[0] = p;
    e->d.n_classes = (int32_t)e->classes.size();
    e->classes_dirty = true;
    e->d.pair_variant = pair_variant_for(e, e->d.n_live);
    {   // no rider model moves faster than its speed clamp (vehicle.py:1258, 1876, 1905; dynamics.py:1025)
        double vmax = 0;
        for (const csf_params &c : e->classes)
            vmax = std::max({vmax, std::fabs(c.v_max_riding[0]), std::fabs(c.v_max_riding[1]), std::fabs(c.v_max_walk)});
        e->d.bnd_margin = (float)(p.t_s * vmax * 1.01 + 1e-4);
    }
    e->d.back = (int32_t)(1.0 / p.t_s);
    int hl = 4;
    while (hl < e->d.back + 2) hl *= 2;
    e->d.hist_len = hl;
    e->d.ns = 0;
    e->d.model_mask = 0;
    for (const csf_params &c : e->classes) {
        e->d.ns = std::max(e->d.ns, NS_OF[c.model]);            // (several vehicle classes: the widest state; unused ones stay 0)
        e->d.model_mask |= 1 << c.model;
    }
    e->d.has_bike = e->d.model_mask & 1;
}

// ---- the lattice over a large road network (csf_road.hip) -----------------------------------------------------------------
// Worth it when the (2 RG_NEAR + 1)^2 cells around a road user hold a small part of the network.  box: x0, x1, y0, y1 of the
// vertices.
static double road_cell_edge(const csf_engine *e) { return e->knobs.road_cell > 0 ? e->knobs.road_cell : 16.0; }
bool road_grid_wanted(const csf_engine *e, double box[4]) {
    const int64_t nv = (int64_t)e->h_road.size() / 4;
    if (nv <= 0 || e->knobs.road_grid == 0) return false;
    box[0] = box[2] = INFINITY, box[1] = box[3] = -INFINITY;
    for (int64_t k = 0; k < nv; k++) {
        box[0] = std::min(box[0], e->h_road[4 * k]), box[1] = std::max(box[1], e->h_road[4 * k]);
        box[2] = std::min(box[2], e->h_road[4 * k + 1]), box[3] = std::max(box[3], e->h_road[4 * k + 1]);
    }
    if (!(std::isfinite(box[0]) && std::isfinite(box[1]) && std::isfinite(box[2]) && std::isfinite(box[3]))) return false;
    if (e->knobs.road_grid == 1) return true;
    const double w = road_cell_edge(e);
    return nv >= 16384 && ((box[1] - box[0]) / w + 1) * ((box[3] - box[2]) / w + 1) >= 4.0 * (2 * RG_NEAR + 1) * (2 * RG_NEAR + 1);
}

// Sorts the vertices into the cells of the lattice, and - once per vertex set - samples every cell's far field at its
// Chebyshev nodes on the device (launch_road_far) and turns the samples into the coefficients of the interpolant.
int build_road_grid(csf_engine *e, const double box[4]) {
    Dev &d = e->d;
    const int64_t nv = d.nv;
    double w = road_cell_edge(e);
    double gx0 = 0, gy0 = 0;
    int64_t nx = 0, ny = 0;
    // The lattice covers the network and a fixed margin around it - RG_NEAR cells or a tenth of the extent, in steps of eight
    // cells - whatever the road users do: its coefficients are then fitted once per network and survive every re-upload (a
    // road user outside it sums every vertex: correct, and as slow as without a lattice).
    const double lo[2] = {box[0], box[2]}, hi[2] = {box[1], box[3]};
    for (;;) {   // coarser cells while the lattice (or the sampling: cells x 64 x nv pairs) is too large
        const double mx = std::max(RG_NEAR * w, 8 * w * std::ceil(0.1 * (hi[0] - lo[0]) / (8 * w)));
        const double my = std::max(RG_NEAR * w, 8 * w * std::ceil(0.1 * (hi[1] - lo[1]) / (8 * w)));
        gx0 = w * std::floor(((lo[0] - mx) - d.ox) / w);
        gy0 = w * std::floor(((lo[1] - my) - d.oy) / w);
        nx = (int64_t)std::floor((((hi[0] + mx) - d.ox) - gx0) / w) + 1;
        ny = (int64_t)std::floor((((hi[1] + my) - d.oy) - gy0) / w) + 1;
        if (nx <= 30000 && ny <= 30000 && nx * ny <= (1 << 18) && (double)nx * (double)ny * 64.0 * (double)nv <= 4e12) break;
        w *= 2;
    }
    const int64_t ncell = nx * ny;
    std::vector<int32_t> start((size_t)ncell + 1, 0), cell((size_t)nv);
    for (int64_t k = 0; k < nv; k++) {
        const int64_t ix = std::min<int64_t>(nx - 1, std::max<int64_t>(0, (int64_t)std::floor(((e->h_road[4 * k] - d.ox) - gx0) / w)));
        const int64_t iy = std::min<int64_t>(ny - 1, std::max<int64_t>(0, (int64_t)std::floor(((e->h_road[4 * k + 1] - d.oy) - gy0) / w)));
        cell[(size_t)k] = (int32_t)(iy * nx + ix);
        start[(size_t)cell[(size_t)k] + 1]++;
    }
    for (int64_t c = 0; c < ncell; c++) start[(size_t)c + 1] += start[(size_t)c];
    std::vector<float4> gv((size_t)nv);
    std::vector<short2> vc((size_t)nv);
    {
        std::vector<int32_t> at(start.begin(), start.end() - 1);
        for (int64_t k = 0; k < nv; k++) {   // (stable: the vertices of a cell keep their order)
            const int32_t c = cell[(size_t)k];
            const int64_t ix = c % nx, iy = c / nx;
            const size_t o = (size_t)at[(size_t)c]++;
            gv[o] = make_float4((float)(((e->h_road[4 * k] - d.ox) - gx0) - ((double)ix + 0.5) * w),
                                (float)(((e->h_road[4 * k + 1] - d.oy) - gy0) - ((double)iy + 0.5) * w),
                                (float)(-e->h_road[4 * k + 2]), (float)(-0.5 * (e->h_road[4 * k + 3] + 1.0)));
            vc[o] = make_short2((short)ix, (short)iy);
        }
    }
    HIPCHK(e, e->rg_v.reserve((size_t)nv));
    HIPCHK(e, e->rg_start.reserve((size_t)ncell + 1));
    HIPCHK(e, hipMemcpy(e->rg_v.p, gv.data(), gv.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(e->rg_start.p, start.data(), start.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    d.rg_nx = (int32_t)nx, d.rg_ny = (int32_t)ny;
    d.rg_w = (float)w, d.rg_x0 = (float)gx0, d.rg_y0 = (float)gy0;
    d.rg_v = e->rg_v.p, d.rg_start = e->rg_start.p;
    d.rg_by_place = 0;
    const bool same = e->rg_version == e->road_version && e->rg_cell == w && e->rg_gx0 == gx0 && e->rg_gy0 == gy0 &&
                      e->rg_nx == (int)nx && e->rg_ny == (int)ny && e->rg_ox == d.ox && e->rg_oy == d.oy && e->rg_c.p != nullptr;
    if (!same) {
        DevBuf<short2> dvc;
        DevBuf<double> dsm;
        HIPCHK(e, dvc.alloc((size_t)nv));
        HIPCHK(e, dsm.alloc((size_t)ncell * 128));
        HIPCHK(e, hipMemcpy(dvc.p, vc.data(), vc.size() * sizeof(short2), hipMemcpyHostToDevice));
        launch_road_far(d, dvc.p, dsm.p, e->main);
        HIPCHK(e, hipGetLastError());
        HIPCHK(e, hipStreamSynchronize(e->main));
        std::vector<double> sm((size_t)ncell * 128);
        HIPCHK(e, hipMemcpy(sm.data(), dsm.p, sm.size() * sizeof(double), hipMemcpyDeviceToHost));
        dvc.release();
        dsm.release();
        // c_ab = k_a k_b (2/8)^2 sum_ij f(x_i, y_j) cos(a pi (i + 1/2) / 8) cos(b pi (j + 1/2) / 8), k_0 = 1/2
        double cs[RG_NODES][RG_NODES];
        for (int a = 0; a < RG_NODES; a++)
            for (int i = 0; i < RG_NODES; i++) cs[a][i] = std::cos(a * 3.14159265358979323846 * (i + 0.5) / RG_NODES);
        std::vector<float> co((size_t)ncell * 128);
        for (int64_t c = 0; c < ncell; c++)
            for (int comp = 0; comp < 2; comp++)
                for (int a = 0; a < RG_NODES; a++)
                    for (int b = 0; b < RG_NODES; b++) {
                        double acc = 0;
                        for (int i = 0; i < RG_NODES; i++)
                            for (int j = 0; j < RG_NODES; j++) acc += sm[((size_t)c * 64 + (size_t)(8 * i + j)) * 2 + (size_t)comp] * cs[a][i] * cs[b][j];
                        acc *= (a ? 1.0 : 0.5) * (b ? 1.0 : 0.5) * (2.0 / RG_NODES) * (2.0 / RG_NODES);
                        co[((size_t)c * 2 + (size_t)comp) * 64 + (size_t)(8 * a + b)] = (float)acc;
                    }
        HIPCHK(e, e->rg_c.reserve(co.size()));
        HIPCHK(e, hipMemcpy(e->rg_c.p, co.data(), co.size() * sizeof(float), hipMemcpyHostToDevice));
        e->rg_version = e->road_version;
        e->rg_cell = w, e->rg_gx0 = gx0, e->rg_gy0 = gy0, e->rg_nx = (int)nx, e->rg_ny = (int)ny, e->rg_ox = d.ox, e->rg_oy = d.oy;
    }
    d.rg_c = e->rg_c.p;
    return CSF_OK;
}

int alloc_all(csf_engine *e) {
    const size_t cap = (size_t)e->cap;
    const size_t hl = (size_t)e->d.hist_len;
    HIPCHK(e, e->s.alloc(STATE_ROWS * cap));
    HIPCHK(e, e->vdes.alloc(cap));
    HIPCHK(e, e->qbeg.alloc(cap));
    HIPCHK(e, e->qlen.alloc(cap));
    HIPCHK(e, e->alive.alloc(cap));
    HIPCHK(e, e->order_dev.alloc(cap));
    HIPCHK(e, e->ptr.alloc(cap));
    HIPCHK(e, e->znav.alloc(cap));
    HIPCHK(e, e->znp.alloc(3 * cap));
    HIPCHK(e, e->ti.alloc(cap));
    HIPCHK(e, e->hx.alloc(hl * cap));
    HIPCHK(e, e->hy.alloc(hl * cap));
    HIPCHK(e, e->lti.alloc(5 * cap));
    HIPCHK(e, e->zrid.alloc(cap));
    HIPCHK(e, e->dgood.alloc(cap));
    HIPCHK(e, e->ppsi.alloc(cap));
    HIPCHK(e, e->F.alloc(6 * cap));
    HIPCHK(e, e->status.alloc(cap));
    HIPCHK(e, e->part.alloc((size_t)MAX_SPLIT * cap));
    HIPCHK(e, e->froad.alloc(cap));
    // records: room for the per-rank padding of an 8-way shard
    size_t nrec = (cap + 64 * 64 + 63) / 64 * 64;
    HIPCHK(e, e->rec.alloc(nrec));
    HIPCHK(e, e->rec2.alloc(nrec));
    {   // the last record of the arrays is no slot's: a sentinel for ever (padding of the class-segmented order)
        const float4 sent = make_float4(1e15f, 1e15f, 1.0f, 0.0f);
        const float2 sent2 = make_float2(0.0f, 1.0f);
        HIPCHK(e, hipMemcpy(e->rec.p + nrec - 1, &sent, sizeof sent, hipMemcpyHostToDevice));
        HIPCHK(e, hipMemcpy(e->rec2.p + nrec - 1, &sent2, sizeof sent2, hipMemcpyHostToDevice));
        e->sent_slot = (int32_t)(nrec - 1);
    }
    HIPCHK(e, e->recs2.alloc(nrec));
    HIPCHK(e, e->perm.alloc(nrec));
    HIPCHK(e, e->pos.alloc(nrec));
    HIPCHK(e, e->recs.alloc(nrec));
    HIPCHK(e, e->recg.alloc(nrec));
    HIPCHK(e, e->recb.alloc(nrec));
    if (g_poison) {   // records that no slot owns: NaN instead of (0, 0, 0, 0), which is a road user at the scene's origin
        const float4 sent = make_float4(1e15f, 1e15f, 1.0f, 0.0f);
        const float2 sent2 = make_float2(0.0f, 1.0f);
        HIPCHK(e, e->rec.poison_from(0)); HIPCHK(e, e->rec2.poison_from(0)); HIPCHK(e, e->recs.poison_from(0));
        HIPCHK(e, e->recs2.poison_from(0)); HIPCHK(e, e->recg.poison_from(0)); HIPCHK(e, e->recb.poison_from(0));
        HIPCHK(e, hipMemcpy(e->rec.p + nrec - 1, &sent, sizeof sent, hipMemcpyHostToDevice));
        HIPCHK(e, hipMemcpy(e->rec2.p + nrec - 1, &sent2, sizeof sent2, hipMemcpyHostToDevice));
    }
    HIPCHK(e, e->borg.alloc(nrec / 64 + 1));
    HIPCHK(e, e->bnd.alloc(nrec / 64));
    HIPCHK(e, e->bnd2.alloc(nrec / 64));
    HIPCHK(e, e->rorg.alloc(nrec));
    HIPCHK(e, e->reclo.alloc(nrec));
    HIPCHK(e, e->sort_vals.alloc(nrec));
    HIPCHK(e, e->rlist.alloc(nrec));
    HIPCHK(e, e->sort_keys.alloc(nrec));
    HIPCHK(e, e->sort_keys_out.alloc(nrec));
    HIPCHK(e, e->sort_tmp.alloc(bin_temp_bytes((int64_t)nrec) + 256));
    e->h_s.assign(STATE_ROWS * cap, 0.0);
    e->h_vdes.assign(cap, 0.0);
    e->h_znp.assign(3 * cap, 0.0);
    e->h_hx.assign(hl * cap, 0.0);
    e->h_hy.assign(hl * cap, 0.0);
    e->h_lti.assign(5 * cap, 0.0);
    e->h_ppsi.assign(cap, 0.0);
    e->h_F.assign(6 * cap, 0.0);
    e->h_ptr.assign(cap, 0);
    e->h_ti.assign(cap, 0);
    e->h_dgood.assign(cap, 0);
    e->h_znav.assign(cap, 0);
    e->h_zrid.assign(cap, 0);
    e->h_status.assign(cap, 0);
    e->h_q.assign(cap, {});
    e->h_script.assign(cap, {});
    HIPCHK(e, e->sbeg.alloc(cap));
    HIPCHK(e, e->slen.alloc(cap));
    e->h_alive.assign(cap, 0);
    e->h_cls.assign(cap, 0);
    HIPCHK(e, e->cls.alloc(cap));
    HIPCHK(e, e->ticket.alloc(1));
    HIPCHK(e, e->edge.alloc(EDGE_CAP));
    HIPCHK(e, e->edge_n.alloc(2));
    HIPCHK(e, e->edge_head.alloc(cap));
    e->sidx.assign(cap, csf_engine::SlotIdx());
    if (!e->bound_pin) HIPCHK(e, hipHostMalloc((void **)&e->bound_pin, 2 * (size_t)cap * sizeof(double), hipHostMallocDefault));
    e->dev_alive.assign(cap, 0);
    Dev &d = e->d;
    d.cap = (int64_t)cap;
    d.s = e->s.p;
    d.vdes = e->vdes.p;
    d.qbeg = e->qbeg.p;
    d.qlen = e->qlen.p;
    d.alive = e->alive.p;
    d.order = nullptr;
    d.ptr = e->ptr.p;
    d.znav = e->znav.p;
    d.znp = e->znp.p;
    d.ti = e->ti.p;
    d.hx = e->hx.p;
    d.hy = e->hy.p;
    d.lti = e->lti.p;
    d.zrid = e->zrid.p;
    d.dgood = e->dgood.p;
    d.ppsi = e->ppsi.p;
    d.F = e->F.p;
    d.F_rows = 6;
    d.status = e->status.p;
    d.part = e->part.p;
    d.froad = e->froad.p;
    d.rec = e->rec.p;
    d.rec2 = e->rec2.p;
    d.recs2 = e->recs2.p;
    d.perm = e->perm.p;
    d.pos = e->pos.p;
    d.recs = e->recs.p;
    d.recg = e->recg.p;
    d.rec_w = d.rec;
    d.recg_w = d.recg;
    d.rec2_w = d.rec2;
    d.recs_w = d.recs;
    d.chase_cnt = nullptr;
    d.chase_misc = nullptr;
    d.chase_round = 0;
    d.chase_gate = 0;
    d.chase_clock = nullptr;
    d.chase_slot = 0;
    d.src64 = e->s.p;
    d.src64_w = nullptr;
    d.mid_group = 0;
    d.recb = e->recb.p;
    d.borg = e->borg.p;
    d.bnd = e->bnd.p;
    d.bnd_next = e->bnd2.p;
    d.rorg = e->rorg.p;
    d.reclo = e->reclo.p;
    d.edge = e->edge.p;
    d.edge_n = e->edge_n.p;
    d.near_dropped = e->edge_n.p + 1;
    d.edge_head = e->edge_head.p;
    d.atrace = nullptr;
    d.snap = nullptr;
    if (!e->knobs.trace_agent.empty()) {   // 8 words per wave of 64 road users
        HIPCHK(e, e->atrace.alloc(8 * ((size_t)cap / 64 + 2) + 16 * ((size_t)cap / 4 + 2)));   // (csf_mid.hip: 16 words per workgroup)
        d.atrace = e->atrace.p;
    }
    d.trace = nullptr;
    if (!e->knobs.trace_blocks.empty()) {  // 3 words per wave: the grid is at most (cap/16) x MAX_SPLIT workgroups of 4 waves
        e->trace_words = 3 * 4 * ((size_t)cap / 16 + 1) * MAX_SPLIT;
        HIPCHK(e, e->trace.alloc(e->trace_words));
        HIPCHK(e, hipMemset(e->trace.p, 0, e->trace_words * sizeof(uint64_t)));
        d.trace = e->trace.p;
    }
    return CSF_OK;
}

constexpr int64_t TAIL_SLOTS = 4096;   // sentinel slots kept behind a binned population (csf_create adds them to the capacity)
void set_chunks(csf_engine *e);

void set_shard(csf_engine *e) {
    Dev &d = e->d;
    d.keep_lo = (e->world > 1 || e->loopback || e->nccl != nullptr) ? 1 : 0;   // (csf_dev.h: reclo)
    d.xbuf = nullptr;
    if (d.keep_lo) {
        // (set_shard has no error path: a failed allocation leaves the pointer NULL, and the exchange itself reports it)
        if (e->xbuf.n < 2 * e->rec.n && e->xbuf.alloc(2 * e->rec.n) != hipSuccess) e->xbuf.release();
        d.xbuf = e->xbuf.p;
    }
    if (e->world <= 1) {
        d.lo = 0;
        d.hi = d.n;
        // (sentinel slots behind the population take the arrivals between two re-binnings, csf_add_agents)
        d.n_pad = (std::min<int64_t>(e->cap, d.n + (e->loopback ? 0 : TAIL_SLOTS)) + 63) / 64 * 64;
    } else {
        // (several parameter sets: the class-segmented order starts every set's run at a multiple of 64 - room for that)
        const int64_t nk = (int64_t)e->classes.size(), seg_pad = nk > 1 && nk <= 16 ? 64 * nk : 0;
        int64_t shard = (d.n + seg_pad + e->world - 1) / e->world;
        shard = (shard + 63) / 64 * 64;
        d.lo = std::min<int64_t>(d.n, (int64_t)e->rank * shard);
        d.hi = std::min<int64_t>(d.n, d.lo + shard);
        d.n_pad = shard * e->world;
    }
    if (e->knobs.fake_world > 1) {  // timing aid (CSF_FAKE_SHARD=r/w): only rank r's receiver block of w is computed
        const int fr = e->knobs.fake_rank, fw = e->knobs.fake_world;   // (no communicator, other blocks' records go stale)
        int64_t shard = ((d.n + fw - 1) / fw + 63) / 64 * 64;
        d.lo = std::min<int64_t>(d.n, (int64_t)fr * shard);
        d.hi = std::min<int64_t>(d.n, d.lo + shard);
    }
    d.n_src = d.n_pad;          // every place of the source order may hold a road user (rebin() knows better)
    e->tail_tracked = false;
    set_chunks(e);
}

// the source chunks of the pair kernel's grid, for the d.n_src places of the source order that can hold road users
void set_chunks(csf_engine *e) {
    Dev &d = e->d;
    int64_t nloc = d.hi - d.lo;
    int64_t blocks = (nloc + 15) / 16;
    int64_t units = std::max<int64_t>(1, d.n_src / 64);
    // Many more workgroups than the chip holds at once: receivers see very different numbers of sources (field
    // of view, position in the scene), so the hardware's dynamic workgroup dispatch is the load balancer.  Chunks
    // of one LDS tile (1024 sources) measured best for 1-, 2-, 4- and 8-way shards of N = 16 384.
    int64_t split;
    if (d.n_src >= 16384) {
        split = blocks > 0 ? (16384 + blocks - 1) / blocks : 1;
        split = std::min<int64_t>(split, d.n_src / 1024);
    } else {
        split = blocks > 0 ? (1024 + blocks - 1) / blocks : 1;
    }
    split = std::max<int64_t>(1, std::min<int64_t>({split, (int64_t)MAX_SPLIT, units}));
    // large populations run the far-tile-skipping variant (rebin: recv_binned): most tiles of a chunk are
    // skipped unloaded, and longer chunks amortise the workgroup's start-up (config 4: 7.3 ms at 64 chunks, 6.2 at 8-16)
    if (d.n_src >= 65536 && d.p.model != CSF_BICYCLE) split = std::min<int64_t>(split, 16);
    if (e->knobs.nsplit > 0) split = std::max<int64_t>(1, std::min<int64_t>({(int64_t)e->knobs.nsplit, (int64_t)MAX_SPLIT, units}));
    // no empty chunk: with per = ceil(units / split) units per chunk only ceil(units / per) chunks hold sources (n = 1040:
    // 17 units, split 16 -> per 2 -> 9 chunks).  A workgroup of an empty chunk would leave its slot of d.part untouched,
    // and the combine phase would add whatever an earlier population layout left there.
    int64_t per = (units + split - 1) / split;
    // a chunk just over one LDS tile (16 batches) would load a second, nearly empty tile in every workgroup: the few
    // batches of arrivals behind a population that filled whole tiles get a chunk - and workgroups - of their own
    if (per > 16 && per < 32 && (units + 15) / 16 <= MAX_SPLIT && e->knobs.nsplit <= 0) per = 16;
    // 32 receivers per workgroup where that still leaves thousands of workgroups (N = 16 384 unsharded: 8192): measured
    // better from 8192 receivers up, worse for the few workgroups of small populations and 4-way shards
    d.rpb = (nloc >= 8192 || (nloc >= 4096 && d.n_src >= 16384)) ? 32 : 16;   // (4-way shard of 16 384: 38.9 -> 36.6 us with the wide workgroups below)
    if (e->knobs.rpb > 0) d.rpb = e->knobs.rpb == 32 ? 32 : e->knobs.rpb == 8 ? 8 : 16;
    // ... and then workgroups of 8 waves on tiles of 2048 sources (csf_pair.hip: CW); below 65 536 places in chunks of 32 batches
    // (with 16 receivers for the shards of a large population: 8-way shard of 16 384 26.3 -> 25.1 us; unsharded 115 against 101)
    d.wide = (d.rpb == 32 || (d.rpb == 16 && (e->knobs.wide > 0 || d.n_src >= 16384))) && d.p.model != CSF_BICYCLE &&
             (e->knobs.nsplit <= 0 || d.n_src >= 65536) && (e->knobs.wide >= 0 ? e->knobs.wide != 0 : true);
    if (d.wide && d.n_src < 65536) per = 32;
    split = (units + per - 1) / per;
    d.n_split = (int32_t)split;
    d.chunk_units = (int32_t)per;
    d.dyn_recv = 1;   // receivers handed to the waves of a workgroup one at a time (csf_pair.hip, DYN; 0: four per wave)
    if (e->knobs.dyn_recv >= 0) d.dyn_recv = e->knobs.dyn_recv != 0;
}

constexpr int64_t BIN_MIN_AGENTS = 1024;

// Which pair kernel (csf_pair.hip: launch_pair): 0 the cull-first kernel (binned records from BIN_MIN_AGENTS road users),
// 1 the plain all-pairs kernel, 2 cull-first without binning.  Below ~3 000 road users the plain kernel is the faster
// one - no classification, no queue, no re-binning launches: 14.1 against 20.7 us per tick at 1 024 TwoDBicycle, 11.1
// against 13.2 at 64, level at ~3 000, 41 against 32 at 4 096 (tools/variant_by_n.py; the Bicycle field crosses over at
// the same size) - and small populations are what the reference itself runs.  CSF_PAIR_VARIANT overrides.
constexpr int64_t PLAIN_BELOW = 3072;
int32_t pair_variant_for(const csf_engine *e, int64_t n) {
    return e->knobs.pair_variant >= 0 ? e->knobs.pair_variant : (n < PLAIN_BELOW ? 1 : 0);
}

// ---- holes (csf_engine::HoleIndex) -----------------------------------------------------------------------------------
// after a re-binning: ask for the places and the circles (two copies and an event behind the re-binning's launches)
static int holes_request(csf_engine *e) {
    csf_engine::HoleIndex &h = e->holes;
    const Dev &d = e->d;
    h.ready = h.pending = false;
    for (auto &c : h.cell_tab) c.n = 0;
    if (!e->knobs.hole_reuse || !e->tail_tracked || !d.classify || d.n_pad <= 0) return CSF_OK;
    const size_t np = (size_t)d.n_pad, nb = np / 64;
    if (h.pos_n < np) {
        if (h.pos) HIPCHK(e, hipHostFree(h.pos));
        h.pos = nullptr;
        HIPCHK(e, hipHostMalloc((void **)&h.pos, np * sizeof(int32_t), hipHostMallocDefault));
        h.pos_n = np;
    }
    if (h.bnd_n < nb) {
        if (h.bnd) HIPCHK(e, hipHostFree(h.bnd));
        h.bnd = nullptr;
        HIPCHK(e, hipHostMalloc((void **)&h.bnd, nb * sizeof(float4), hipHostMallocDefault));
        h.bnd_n = nb;
    }
    if (!h.ev) HIPCHK(e, hipEventCreateWithFlags(&h.ev, hipEventDisableTiming));
    HIPCHK(e, hipMemcpyAsync(h.pos, d.pos, np * sizeof(int32_t), hipMemcpyDeviceToHost, e->main));
    HIPCHK(e, hipMemcpyAsync(h.bnd, d.bnd, nb * sizeof(float4), hipMemcpyDeviceToHost, e->main));
    HIPCHK(e, hipEventRecord(h.ev, e->main));
    h.places = (e->live_at_rebin + 63) / 64 * 64;
    h.ox = d.ox, h.oy = d.oy;
    h.pending = true;
    return CSF_OK;
}

// the lattice, once the read-back has landed (the first population call after a re-binning waits for it, if at all)
static bool holes_ready(csf_engine *e) {
    csf_engine::HoleIndex &h = e->holes;
    if (h.ready) return true;
    if (!h.pending) return false;
    if (hipEventSynchronize(h.ev) != hipSuccess) return h.pending = false;
    h.pending = false;
    const int64_t nb = h.places / 64;
    h.hpos.assign(h.pos, h.pos + h.pos_n);
    h.hbnd.assign(h.bnd, h.bnd + h.bnd_n);
    std::vector<float> radii;
    double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
    for (int64_t b = 0; b < nb; b++) {
        const float4 c = h.hbnd[(size_t)b];
        if (!(c.z < 1e6f) || !std::isfinite(c.x) || !std::isfinite(c.y)) continue;       // (a batch of sentinels)
        radii.push_back(c.z);
        x0 = std::min(x0, (double)c.x), x1 = std::max(x1, (double)c.x), y0 = std::min(y0, (double)c.y), y1 = std::max(y1, (double)c.y);
    }
    if (radii.size() < 4) return false;
    std::nth_element(radii.begin(), radii.begin() + radii.size() / 2, radii.end());
    const double reach = e->knobs.hole_dist * (double)radii[radii.size() / 2];
    if (!(reach > 0.0)) return false;
    h.reach2 = reach * reach;
    h.cell = reach;
    h.x0 = x0 - reach, h.y0 = y0 - reach;
    h.nx = (int)std::min(512.0, std::ceil((x1 - x0 + 2 * reach) / h.cell) + 1);
    h.ny = (int)std::min(512.0, std::ceil((y1 - y0 + 2 * reach) / h.cell) + 1);
    h.cell = std::max({h.cell, (x1 - x0 + 2 * reach) / (h.nx - 1), (y1 - y0 + 2 * reach) / (h.ny - 1)});
    h.inv_cell = 1.0 / h.cell;
    h.cell_tab.assign((size_t)h.nx * (size_t)h.ny, csf_engine::HoleIndex::Cell{});
    return h.ready = true;
}

// a road user has left slot a: if the slot had a place in a real batch, its hole can be found by the batch's centre
static void holes_add(csf_engine *e, int32_t a) {
    csf_engine::HoleIndex &h = e->holes;
    if (!holes_ready(e) || (size_t)a >= h.hpos.size()) return;
    const int64_t place = h.hpos[(size_t)a];
    if (place < 0 || place >= h.places) return;                    // (it arrived after the re-binning: a place of the tail)
    const float4 c = h.hbnd[(size_t)(place >> 6)];
    if (!(c.z < 1e6f)) return;
    const int ix = (int)(((double)c.x - h.x0) * h.inv_cell), iy = (int)(((double)c.y - h.y0) * h.inv_cell);
    if (ix < 0 || iy < 0 || ix >= h.nx || iy >= h.ny) return;
    csf_engine::HoleIndex::Cell &cl = h.cell_tab[(size_t)iy * h.nx + ix];
    if (cl.n >= csf_engine::HoleIndex::CELL_CAP) return;          // (a crowded cell: this one is found by the re-binning only)
    cl.e[cl.n++] = {a, c.x, c.y};
}

// the hole nearest to (x, y) whose batch's centre is within reach; -1: none
static int32_t holes_take(csf_engine *e, double x, double y) {
    csf_engine::HoleIndex &h = e->holes;
    if (!h.ready || e->free_recent.empty()) return -1;
    const float sx = (float)(x - h.ox), sy = (float)(y - h.oy);    // scene coordinates, as the circles
    const int ix = (int)std::floor(((double)sx - h.x0) * h.inv_cell), iy = (int)std::floor(((double)sy - h.y0) * h.inv_cell);
    const int jx0 = std::max(0, ix - 1), jx1 = std::min(h.nx - 1, ix + 1), jy0 = std::max(0, iy - 1), jy1 = std::min(h.ny - 1, iy + 1);
    for (;;) {
        float best = (float)h.reach2;
        csf_engine::HoleIndex::Cell *bc = nullptr;
        int bi = -1;
        auto look = [&](csf_engine::HoleIndex::Cell &cl) {
            for (int i = 0; i < cl.n; i++) {
                const float dx = cl.e[i].x - sx, dy = cl.e[i].y - sy, d2 = dx * dx + dy * dy;
                if (d2 < best) best = d2, bc = &cl, bi = i;
            }
        };
        // its own cell first: a hole within reach there is as good as any (its batch's circle holds the arrival, or nearly)
        if (ix >= 0 && iy >= 0 && ix < h.nx && iy < h.ny) look(h.cell_tab[(size_t)iy * h.nx + ix]);
        if (bi < 0)                                                // (nobody there: the eight cells around)
            for (int jy = jy0; jy <= jy1; jy++)
                for (int jx = jx0; jx <= jx1; jx++)
                    if (jx != ix || jy != iy) look(h.cell_tab[(size_t)jy * h.nx + jx]);
        if (bi < 0) return -1;
        const int32_t a = bc->e[bi].slot;
        bc->e[bi] = bc->e[--bc->n];
        const int32_t at = e->sidx[(size_t)a].recent;
        if (at < 0) continue;                                      // (handed out some other way since: look again)
        const int32_t last = e->free_recent.back();                // out of free_recent as well
        e->free_recent[(size_t)at] = last;
        e->sidx[(size_t)last].recent = at;
        e->free_recent.pop_back();
        e->sidx[(size_t)a].recent = -1;
        h.taken++;
        return a;
    }
}

// (re)build the spatially binned order of the records; decides whether batches are classified this tick
int rebin(csf_engine *e) {
    Dev &d = e->d;
    e->mid_synced = false;     // (records re-expressed, perhaps a new sentinel tail: csf_mid.hip's other halves are stale)
    if (e->world <= 1 && !e->loopback && d.n_pad < e->cap && d.n + TAIL_SLOTS / 2 > d.n_pad) {   // fresh slots are running out
        set_shard(e);
        launch_records(d, e->main);     // sentinels in the new tail (the records of the road users are rewritten as they are)
    }
    d.pair_variant = pair_variant_for(e, d.n_live);
    // Several parameter sets: up to 16 of them get the class-segmented order (a rank of a sharded run too: the order is one of
    // the SOURCES, which every rank holds in full, and the same on every rank) - the set leads the
    // sort key, every set becomes a run of places that starts at a multiple of 64, and the pair term is one launch of the
    // culling kernel per run with that set's constants, far-field radius and field (launch_pair_all).  Otherwise the
    // plain kernel looks every source's set up (csf_pair.hip: HET).
    e->segs.clear();
    e->h_segtab.clear();
    e->seg_total_by = 0;
    d.seg_keys = 0;
    d.src_beg = 0;
    d.part_base = 0;
    // It pays from ~2 048 road users per set (a launch per set has its own start-up): four sets at N = 16 384 199 us per
    // tick against 387 us, at 8 192 104 against 119, at 4 096 68 against 41 (tools/hetero_rate.py; CSF_SEGMENTS=1 forces
    // it from 1 024 road users, 0 switches it off).
    const int seg_env = e->knobs.segments;
    bool seg = d.n_classes > 1 && d.n_classes <= 16 && d.pair_variant == 0 && d.n >= BIN_MIN_AGENTS &&
               e->class_kappa.size() == e->classes.size() &&
               (seg_env >= 0 ? seg_env != 0 : d.n_live >= 2048 * (int64_t)d.n_classes);
    SegTable tab{};
    if (seg) {
        std::vector<int64_t> count((size_t)d.n_classes, 0);
        for (int32_t a : e->order) count[e->h_cls[(size_t)a]]++;
        int64_t sorted = 0, place = 0, units_total = 0;
        tab.n = d.n_classes;
        tab.sent_slot = e->sent_slot;
        for (int c = 0; c < d.n_classes; c++) {
            tab.sorted_beg[c] = sorted;
            tab.place_beg[c] = place;
            sorted += count[(size_t)c];
            place += (count[(size_t)c] + 63) / 64 * 64;
        }
        tab.sorted_beg[d.n_classes] = sorted;
        units_total = place / 64;
        if (place > d.n_pad || sorted != d.n_live) seg = false;      // (no room for the padding: the plain kernel)
        if (seg) {
            const int64_t per_min = std::max<int64_t>(d.wide ? 32 : 16, (units_total + (MAX_SPLIT - d.n_classes) - 1) / std::max(1, MAX_SPLIT - d.n_classes));
            int32_t slots = 0;
            for (int c = 0; c < d.n_classes; c++) {
                const int64_t units = (count[(size_t)c] + 63) / 64;
                PairConsts pc;
                derive_pair_consts(e->classes[(size_t)c], pc, e->knobs.rnear);
                pc.p2r = d.pc.p2r;
                set_far_consts(e->knobs, e->classes[(size_t)c], e->class_kappa[(size_t)c], d.n, pc);
                if (units == 0 || pc.f0_zero) continue;              // nobody of this set, or a set whose field is zero
                const int64_t chunks = (units + per_min - 1) / per_min;
                csf_engine::Segment sg;
                sg.cls = c;
                sg.beg = tab.place_beg[c];
                sg.end = sg.beg + units * 64;
                sg.chunk_units = (int32_t)((units + chunks - 1) / chunks);
                sg.n_split = (int32_t)((units + sg.chunk_units - 1) / sg.chunk_units);
                sg.part_base = slots;
                sg.pc = pc;
                slots += sg.n_split;
                e->segs.push_back(sg);
            }
            if (slots > MAX_SPLIT) seg = false, e->segs.clear();
            else {
                d.seg_keys = 1;
                d.n_split = std::max(1, slots);                       // what the per-agent kernel sums
                d.n_src = place;
            }
            // the runs whose field is the TwoD one, as the segmented grid reads them
            e->h_segtab.clear();
            e->seg_total_by = 0;
            for (const csf_engine::Segment &sg : e->segs) {
                if (e->classes[(size_t)sg.cls].model == CSF_BICYCLE) continue;
                SegDev t{};
                t.pc = sg.pc;
                t.hfov = e->classes[(size_t)sg.cls].hfov;
                t.src_beg = sg.beg;
                t.n_src = sg.end;
                t.chunk_units = sg.chunk_units;
                t.n_split = sg.n_split;
                t.part_base = sg.part_base;
                t.first_by = e->seg_total_by;
                e->seg_total_by += sg.n_split;
                e->h_segtab.push_back(t);
            }
            bool same_reach = true;
            for (const SegDev &t : e->h_segtab) same_reach = same_reach && t.pc.reach != 0;
            if (e->h_segtab.size() < 2 || !same_reach || e->h_segtab.size() > 16) e->seg_total_by = 0;   // (one run: nothing to merge)
            else {
                HIPCHK(e, e->segtab.reserve(16));
                HIPCHK(e, hipMemcpyAsync(e->segtab.p, e->h_segtab.data(), e->h_segtab.size() * sizeof(SegDev), hipMemcpyHostToDevice, e->main));
            }
        }
    }
    const bool binned = d.pair_variant == 0 && d.n >= BIN_MIN_AGENTS && (d.n_classes == 1 || seg);
    d.classify = binned;
    update_far_radius(e);
    const bool track = binned && !seg && e->world <= 1 && !e->loopback;
    if (seg) {
        Dev ds = d;
        ds.perm = e->rlist.p;                                        // the sorted slots, before the runs are moved apart
        int rc = launch_rebin(ds, e->sort_keys.p, e->sort_keys_out.p, e->sort_vals.p, e->sort_tmp.p, e->sort_tmp.n, e->main);
        if (rc != 0) return fail(e, CSF_E_DEVICE, "radix sort of the record bins failed (%d)", rc);
        launch_segment_perm(d, e->rlist.p, tab, e->main);
    } else if (binned) {
        int rc = launch_rebin(d, e->sort_keys.p, e->sort_keys_out.p, e->sort_vals.p, e->sort_tmp.p, e->sort_tmp.n, e->main);
        if (rc != 0) return fail(e, CSF_E_DEVICE, "radix sort of the record bins failed (%d)", rc);
    } else {
        launch_identity_perm(d, e->main);
    }
    // Every precise record gets a new origin - where the road user is now - and is re-expressed relative to it; pos[],
    // the binned copy of the records in scene coordinates and the circles (csf_bin.hip: rebase_kernel).  The binned
    // copy is maintained by the agent kernel where every record is local, rebuilt from the gathered records before every
    // pair launch of a sharded run (enqueue_tick).
    d.recs_valid = binned;
    d.rebase_from_state = e->state_all_current;
    launch_rebase(d, e->main);
    {   // receivers in binned order + far-tile skipping, where there are enough tiles for it to pay.  A rank that owns
        // an index block [lo, hi) takes ITS receivers in binned order: their positions, sorted
        const int ov = e->knobs.recv_binned;
        const bool whole = d.lo == 0 && d.hi == d.n;
        d.recv_binned = binned && d.recs_valid && std::isfinite(d.pc.rfar) && (ov >= 0 ? ov != 0 : d.n_pad >= 65536);
        d.rlist = nullptr;
        if (d.recv_binned && !whole) {
            int rc = launch_receiver_list(d, e->sort_keys.p, e->rlist.p, e->sort_tmp.p, e->sort_tmp.n, e->main);
            if (rc != 0) return fail(e, CSF_E_DEVICE, "radix sort of the receiver positions failed (%d)", rc);
            d.rlist = e->rlist.p;
        }
    }
    // Where the sentinels went: the sort is stable and their key is the largest, so the road users fill the places
    // [0, n_live) and the free slots follow in ascending slot order.  Handing the free slots out in that order
    // (csf_add_agents) keeps the places that can hold a road user a prefix of the order, and the pair kernel's source
    // chunks end there (d.n_src) instead of at n_pad.  Slots retired since the last re-binning are sentinels from now on.
    e->tail_tracked = track;
    for (int32_t a : e->free_recent) e->sidx[(size_t)a].recent = -1;   // (every hole goes to the tail: the order has none)
    e->free_tail.insert(e->free_tail.end(), e->free_recent.begin(), e->free_recent.end());
    e->free_recent.clear();
    std::sort(e->free_tail.begin(), e->free_tail.end(), std::greater<int32_t>());
    e->live_at_rebin = d.n_live;
    e->tail_used = 0;
    e->tail_flushed = 0;
    if (!seg) {
        const int64_t n_src = e->tail_tracked ? std::max<int64_t>(64, (e->live_at_rebin + 63) / 64 * 64)
                              : (e->world <= 1 && !e->loopback && !binned ? std::max<int64_t>(64, (d.n + 63) / 64 * 64) : d.n_pad);
        d.n_src = std::min(n_src, d.n_pad);
        set_chunks(e);                                               // (also after a segmented period: n_split was the segments')
    }
    d.clist = nullptr;
    d.ccount = nullptr;
    if (d.recv_binned && e->knobs.clist != 0) {   // which tiles can matter to which receiver group until the next re-binning
        d.clist_tile = d.wide ? 2048 : 1024;
        d.clist_rpb = d.rpb;
        const int64_t nloc = d.hi - d.lo, groups = (nloc + d.rpb - 1) / d.rpb;
        const int64_t ntiles = (d.n_src + d.clist_tile - 1) / d.clist_tile;
        if (groups > 0 && ntiles > 0 && ntiles < 65536) {
            HIPCHK(e, e->tcirc.reserve((size_t)ntiles + 1));
            HIPCHK(e, e->clist.reserve((size_t)groups * CLIST_MAX));
            HIPCHK(e, e->ccount.reserve((size_t)groups));
            // tiles that hold road users as of now are listed; the tile the population ends in and the sentinel tail behind it
            // (where arrivals appear) are always visited
            d.ctail = (int32_t)(e->tail_tracked ? e->live_at_rebin / d.clist_tile : ntiles);
            // both sides move until the lists are rebuilt (REBIN_TICKS ticks; a churn-triggered re-binning comes sooner)
            const float move = (float)(e->knobs.rebin_ticks + 2) * d.bnd_margin + 5e-3f;   // what either side can move until then
            const float reach = d.pc.rfar + 2.0f * move;
            // The far-field bound from the tiles within reach (large populations): R = ln(n / eps) / kappa prices every one of
            // the n sources at the cut - a million road users pay for a radius that holds a few ten thousand.  With the lists
            // in hand: a receiver meets at most m sources (the places of its group's listed tiles and of the tiles always
            // visited), and every source of a tile it does not list is at least the circles' separation away and adds at
            // most f_0 exp(-kappa separation) - `tail` in all.  Leaving out the pairs with rho q / sigma > T among the m
            // omits at most m exp(-T) f_0 more: T = ln(m / (eps - tail)) keeps the total below eps f_0, the same promise
            // (include/csf.h: csf_far_radius) at a smaller radius.  m and tail are maxima over the groups of this device,
            // measured at every re-binning; one read-back of two words, where a tick takes milliseconds.
            const bool tighten = e->knobs.far_tight != 0 && d.pc.reach && e->classes.size() == 1 && e->far_kappa > 0;
            if (tighten) {
                HIPCHK(e, e->far_stat.reserve(2));
                HIPCHK(e, hipMemsetAsync(e->far_stat.p, 0, 2 * sizeof(unsigned), e->main));
            }
            launch_candidate_lists(d, e->tcirc.p, e->clist.p, e->ccount.p, reach, e->main, tighten ? e->far_stat.p : nullptr, (float)e->far_kappa, move);
            d.clist = e->clist.p;
            d.ccount = e->ccount.p;
            if (tighten) {
                // (into the pinned buffer csf_create made for the coordinate bound - cap >= 64 doubles; the stream is waited for
                // right below, so the two uses cannot meet: a runtime's first copy into PAGEABLE memory cost 9 ms in mid-run)
                unsigned st[2] = {0u, 0u};
                if (!e->bound_pin) HIPCHK(e, hipHostMalloc((void **)&e->bound_pin, 2 * (size_t)e->cap * sizeof(double), hipHostMallocDefault));
                HIPCHK(e, hipMemcpyAsync(e->bound_pin, e->far_stat.p, sizeof st, hipMemcpyDeviceToHost, e->main));
                HIPCHK(e, hipStreamSynchronize(e->main));
                std::memcpy(st, e->bound_pin, sizeof st);
                float tail;
                std::memcpy(&tail, &st[1], sizeof tail);
                const double eps = e->knobs.far_eps, met = std::min<double>((double)st[0], (double)d.n) + (double)TAIL_SLOTS;   // (+ arrivals until then)
                e->far_met = met;
                e->far_tail = tail;
                if (st[0] > 0 && std::isfinite(tail) && tail < 0.5 * eps) {
                    const double T = std::log(met / (eps - (double)tail)), T0 = std::log((double)d.n / eps);
                    if (T < T0) {                                    // (the lists were built for the larger radius: still complete)
                        set_far_consts_T(d.p, e->far_kappa, T, d.pc);
                        e->far_T = T;
                    }
                }
            }
        }
    }
    // set_fov_band bounds the coordinates by "where they were at the last upload + a speed clamp's worth per step since": over
    // 1e5 ticks that bound - and with it every rounding band - grows far beyond the scene.  Measure again now and then
    // (one read-back of the positions; where every slot's state is on this device).
    if (e->moves >= 4096 && e->world <= 1 && !e->loopback) e->bound_stale = true;
    e->ticks_since_rebin = 0;
    e->moved_unbinned = 0;
    e->churn = 0;
    e->bounds_fresh = true;                                          // (rebase_kernel wrote the circles of the records as they are)
    return holes_request(e);                                         // (places and circles of this order, for arrivals that take a leaver's slot)
}

// bounding circles for the pair launch that follows; afterwards the circles emitted by that launch become current
int bounds_before_pair(csf_engine *e) {
    Dev &d = e->d;
    // Road users that arrived since the last re-binning sit in the tail batches of the binned order, among each other
    // and far from sorted: every receiver tests those batches lane by lane (measured ~1.3 us per batch and tick at
    // N = 16 384), against ~35 us for the kernels of a re-binning.  With r arrivals per tick the cheapest period is about
    // sqrt(2 * 35 * 64 / (1.3 r)) ticks, i.e. re-bin when ticks x arrivals since the last one reaches ~3500; 3000 to 6000
    // measured alike (profiles/r2_churn_rate.txt; CSF_REBIN_CHURN overrides the constant).
    const int64_t churn_k = e->knobs.rebin_churn;
    const bool rebin_now = e->ticks_since_rebin + e->moved_unbinned >= e->knobs.rebin_ticks || e->ticks_since_rebin * e->churn >= churn_k;
    // Exchange records that arrived behind the last tick (a rank of a sharded run): spread to rec / reclo / rec2 by the copy into
    // binned order that follows the circles - unless this call re-bins (the re-binning reads the records) or no such copy will
    // run (records not binned; the first tick of a binned order, whose copy the re-binning made itself)
    if (e->xbuf_fresh && (rebin_now || !d.recs_valid || e->ticks_since_rebin < 1)) {
        launch_unpack_exchange(d, e->main);
        e->xbuf_fresh = false;
    }
    if (rebin_now) {
        int rc = rebin(e);
        if (rc) return rc;
    }
    if (d.classify && !e->bounds_fresh) launch_bounds(d, e->main);
    e->ticks_since_rebin++;
    e->pair_since_move = true;
    return CSF_OK;
}

void bounds_after_pair(csf_engine *e, bool records_will_move_one_tick) {
    Dev &d = e->d;
    if (!d.classify || d.n_live <= 1) {
        e->bounds_fresh = false;
        return;
    }
    std::swap(d.bnd, d.bnd_next);                   // the launch wrote the next tick's circles
    e->bounds_fresh = records_will_move_one_tick;   // valid only if exactly one integrate follows
}

int flush_pending(csf_engine *e);   // collected population changes -> the device (defined with the population entry points)

// device -> host mirror (needed before a structural change once ticks have run)
int download_all(csf_engine *e) {
    int frc = flush_pending(e);
    if (frc) return frc;
    if (!e->device_ahead) return CSF_OK;
    HIPCHK(e, hipStreamSynchronize(e->main));
    if (e->comm) HIPCHK(e, hipStreamSynchronize(e->comm));
#define D2H(vec, buf) HIPCHK(e, hipMemcpy(vec.data(), buf.p, vec.size() * sizeof(vec[0]), hipMemcpyDeviceToHost))
    D2H(e->h_s, e->s);
    D2H(e->h_znp, e->znp);
    D2H(e->h_hx, e->hx);
    D2H(e->h_hy, e->hy);
    D2H(e->h_lti, e->lti);
    D2H(e->h_ppsi, e->ppsi);
    D2H(e->h_F, e->F);
    D2H(e->h_ptr, e->ptr);
    D2H(e->h_ti, e->ti);
    D2H(e->h_dgood, e->dgood);
    D2H(e->h_znav, e->znav);
    D2H(e->h_zrid, e->zrid);
    D2H(e->h_status, e->status);
#undef D2H
    e->device_ahead = false;
    return CSF_OK;
}

// slots -> population order on the host mirror (a full upload starts from a population without holes)
void compact_host(csf_engine *e) {
    const int64_t n = (int64_t)e->order.size(), cap = e->cap, hl = e->d.hist_len;
    bool identity = e->free_tail.empty() && e->free_recent.empty() && n == e->d.n;
    for (int64_t i = 0; identity && i < n; i++) identity = e->order[(size_t)i] == i;
    if (!identity) {
        auto gather = [&](auto &vec, int64_t comps) {
            auto old = vec;
            for (int64_t c = 0; c < comps; c++)
                for (int64_t i = 0; i < n; i++) vec[(size_t)(c * cap + i)] = old[(size_t)(c * cap + e->order[(size_t)i])];
        };
        gather(e->h_s, STATE_ROWS); gather(e->h_F, 6); gather(e->h_znp, 3); gather(e->h_lti, 5);
        gather(e->h_hx, hl); gather(e->h_hy, hl);
        gather(e->h_vdes, 1); gather(e->h_ppsi, 1); gather(e->h_ptr, 1); gather(e->h_ti, 1); gather(e->h_dgood, 1);
        gather(e->h_znav, 1); gather(e->h_zrid, 1); gather(e->h_status, 1); gather(e->h_cls, 1);
        std::vector<std::vector<double>> q((size_t)cap), sc((size_t)cap);
        for (int64_t i = 0; i < n; i++) q[(size_t)i] = std::move(e->h_q[(size_t)e->order[(size_t)i]]);
        for (int64_t i = 0; i < n; i++) sc[(size_t)i] = std::move(e->h_script[(size_t)e->order[(size_t)i]]);
        e->h_q.swap(q);
        e->h_script.swap(sc);
        for (int64_t i = 0; i < n; i++) e->order[(size_t)i] = (int32_t)i;
        e->free_tail.clear();
        e->free_recent.clear();
        for (csf_engine::SlotIdx &x : e->sidx) x.recent = -1;
    }
    std::fill(e->h_alive.begin(), e->h_alive.end(), (uint8_t)0);
    std::fill(e->h_alive.begin(), e->h_alive.begin() + n, (uint8_t)1);
    std::fill(e->h_cls.begin() + n, e->h_cls.end(), (uint8_t)0);
    e->d.n = n;
    e->d.n_live = n;
    e->order_dirty = true;
}

// destination queues of the live slots -> one slab of rows, every queue contiguous, with room behind them for the queues
// of road users that arrive and for queues that are replaced (those are appended, flush_pending), `extra` rows at least
int upload_queues(csf_engine *e, int64_t extra) {
    Dev &d = e->d;
    std::vector<int64_t> beg((size_t)e->cap, 0);
    std::vector<int32_t> len((size_t)e->cap, 0);
    int64_t rows = 0;
    for (int64_t a = 0; a < d.n; a++) {
        if (!e->h_alive[(size_t)a]) continue;
        beg[(size_t)a] = rows;
        len[(size_t)a] = (int32_t)(e->h_q[(size_t)a].size() / 3);
        rows += len[(size_t)a];
    }
    if ((size_t)(3 * (rows + extra)) > e->q.n || e->q.n == 0) {
        size_t want = (size_t)std::max<int64_t>(3 * (rows + extra) * 2, 3 * 4096);
        HIPCHK(e, e->q.alloc(want));
    }
    // A slab that is rewritten often (compact_slab: arrivals and new routes every tick - each such rewrite waits for the ticks in
    // flight and uploads every live queue, ~1 ms at N = 16 384, and at 5 % of the population replaced per tick it came every 17
    // ticks: 59 us per tick, most of the host's time in csf_set_dest_queue) grows to eight times what is alive: a rewrite every
    // ~140 such ticks.  (Up to 96 MB; beyond, twice the live rows as before.)
    if (e->slab_rewrites >= 2 && (size_t)(3 * (rows + extra)) * 8 > e->q.n && (size_t)(3 * (rows + extra)) * 8 * sizeof(double) <= (96u << 20))
        HIPCHK(e, e->q.alloc((size_t)(3 * (rows + extra)) * 8));
    d.q = e->q.p;
    d.qcap = (int64_t)(e->q.n / 3);
    e->q_top = rows;
    std::vector<double> flat((size_t)(3 * rows), 0.0);
    for (int64_t a = 0; a < d.n; a++) {
        if (!e->h_alive[(size_t)a]) continue;
        const std::vector<double> &qa = e->h_q[(size_t)a];
        std::copy(qa.begin(), qa.end(), flat.begin() + 3 * beg[(size_t)a]);
    }
    if (!flat.empty()) HIPCHK(e, hipMemcpy(e->q.p, flat.data(), flat.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(e->qbeg.p, beg.data(), beg.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(e->qlen.p, len.data(), len.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return CSF_OK;
}

// The slab is full of queues that were replaced or whose road users left: write the live queues afresh (the host holds
// every queue as it was last set; the pointers into them live on the device and stay as they are).
int compact_slab(csf_engine *e, int64_t extra) {
    e->slab_rewrites++;
    int rc = flush_pending(e);
    if (rc) return rc;
    HIPCHK(e, hipStreamSynchronize(e->main));                  // ticks in flight read the old slab
    return upload_queues(e, extra);
}

// the table of parameter sets, what the kernels derive from each row, and the row of every slot
int upload_classes(csf_engine *e) {
    Dev &d = e->d;
    const size_t K = e->classes.size();
    if (e->ptab.n < K) {
        HIPCHK(e, e->ptab.alloc(K));
        HIPCHK(e, e->pctab.alloc(K));
        HIPCHK(e, e->pbtab.alloc(7 * K));
    }
    std::vector<PairConsts> pc(K);
    std::vector<double> pb(7 * K, 0.0);
    e->class_kappa.assign(K, 0.0);
    for (size_t c = 0; c < K; c++) {
        derive_pair_consts(e->classes[c], pc[c], e->knobs.rnear);
        if (K > 1 && K <= 16) e->class_kappa[c] = far_kappa(e->classes[c]);   // (for the launches per set: rebin)
        pc[c].p2r = d.pc.p2r;                                    // (the rule belongs to the intersection: intersection.py:324)
        if (e->classes[c].model == CSF_PLANARBIKE) derive_planarbike(e->classes[c], &pb[7 * c]);
    }
    HIPCHK(e, hipStreamSynchronize(e->main));                  // ticks in flight read the old rows
    HIPCHK(e, hipMemcpy(e->ptab.p, e->classes.data(), K * sizeof(csf_params), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(e->pctab.p, pc.data(), K * sizeof(PairConsts), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(e->pbtab.p, pb.data(), pb.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(e->cls.p, e->h_cls.data(), e->h_cls.size(), hipMemcpyHostToDevice));
    d.ptab = e->ptab.p;
    d.pctab = e->pctab.p;
    d.pbtab = e->pbtab.p;
    d.cls = e->cls.p;
    d.n_classes = (int32_t)K;
    e->classes_dirty = false;
    return CSF_OK;
}

int upload_all(csf_engine *e) {
    if (e->classes_dirty) {
        int rc = upload_classes(e);
        if (rc) return rc;
    }
    if (!e->dirty) return flush_pending(e);
    Dev &d = e->d;
    e->mid_synced = false;
    compact_host(e);
    if (e->classes.size() > 1) HIPCHK(e, hipMemcpy(e->cls.p, e->h_cls.data(), e->h_cls.size(), hipMemcpyHostToDevice));   // (slots moved)
    const int64_t n = d.n;
    int qrc = upload_queues(e, 0);
    if (qrc) return qrc;
    {   // the prescribed trajectories of UncontrolledVehicle road users (vehicle.py:958-960): one slab of rows
        std::vector<int64_t> beg((size_t)e->cap, 0);
        std::vector<int32_t> len((size_t)e->cap, 0);
        std::vector<double> flat;
        for (int64_t a = 0; a < n; a++) {
            beg[(size_t)a] = (int64_t)(flat.size() / 4);
            len[(size_t)a] = (int32_t)(e->h_script[(size_t)a].size() / 4);
            flat.insert(flat.end(), e->h_script[(size_t)a].begin(), e->h_script[(size_t)a].end());
        }
        if (flat.size() > e->script.n) HIPCHK(e, e->script.alloc(flat.size()));
        if (!flat.empty()) HIPCHK(e, hipMemcpy(e->script.p, flat.data(), flat.size() * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(e, hipMemcpy(e->sbeg.p, beg.data(), beg.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        HIPCHK(e, hipMemcpy(e->slen.p, len.data(), len.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        d.script = e->script.p;
        d.sbeg = e->sbeg.p;
        d.slen = e->slen.p;
    }
    HIPCHK(e, hipMemcpy(e->alive.p, e->h_alive.data(), e->h_alive.size(), hipMemcpyHostToDevice));
    e->dev_alive = e->h_alive;
    if (g_poison) {   // the slots behind the population (no road user, no sentinel duty but the record): NaN, not zero
        const double qnan = std::numeric_limits<double>::quiet_NaN();
        auto fill = [&](std::vector<double> &v) {
            const size_t rows = v.size() / (size_t)e->cap;
            for (size_t r = 0; r < rows; r++) std::fill(v.begin() + r * e->cap + n, v.begin() + (r + 1) * e->cap, qnan);
        };
        fill(e->h_s); fill(e->h_vdes); fill(e->h_znp); fill(e->h_hx); fill(e->h_hy); fill(e->h_lti); fill(e->h_ppsi); fill(e->h_F);
    }
#define H2D(vec, buf) HIPCHK(e, hipMemcpy(buf.p, vec.data(), vec.size() * sizeof(vec[0]), hipMemcpyHostToDevice))
    H2D(e->h_s, e->s);
    H2D(e->h_vdes, e->vdes);
    H2D(e->h_znp, e->znp);
    H2D(e->h_hx, e->hx);
    H2D(e->h_hy, e->hy);
    H2D(e->h_lti, e->lti);
    H2D(e->h_ppsi, e->ppsi);
    H2D(e->h_F, e->F);
    H2D(e->h_ptr, e->ptr);
    H2D(e->h_ti, e->ti);
    H2D(e->h_dgood, e->dgood);
    H2D(e->h_znav, e->znav);
    H2D(e->h_zrid, e->zrid);
    H2D(e->h_status, e->status);
#undef H2D
    // origin of the fp32 records: centre of the bounding box of the population (and road)
    if (n > 0) {
        double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
        for (int64_t a = 0; a < n; a++) {
            x0 = std::min(x0, e->h_s[a]);
            x1 = std::max(x1, e->h_s[a]);
            y0 = std::min(y0, e->h_s[e->cap + a]);
            y1 = std::max(y1, e->h_s[e->cap + a]);
        }
        d.ox = 0.5 * (x0 + x1);
        d.oy = 0.5 * (y0 + y1);
        if (!std::isfinite(d.ox)) d.ox = 0;
        if (!std::isfinite(d.oy)) d.oy = 0;
    }
    double rbox[4];
    const bool road_grid = road_grid_wanted(e, rbox);
    if (road_grid) {   // a lattice over the road network (csf_road.hip) is anchored at the origin: the network's centre, which stays
        d.ox = std::nearbyint(0.5 * (rbox[0] + rbox[1]));
        d.oy = std::nearbyint(0.5 * (rbox[2] + rbox[3]));
    }
    {   // the largest coordinate a record can take, relative to that origin (set_fov_band): road users and the prescribed
        // trajectories of UncontrolledVehicles now; arrivals and motion are added as they come
        double cb = 0;
        for (int64_t a = 0; a < n; a++) {
            cb = std::max({cb, std::fabs(e->h_s[a] - d.ox), std::fabs(e->h_s[e->cap + a] - d.oy)});
            const std::vector<double> &sc = e->h_script[(size_t)a];
            for (size_t r = 0; r + 3 < sc.size(); r += 4) cb = std::max({cb, std::fabs(sc[r] - d.ox), std::fabs(sc[r + 1] - d.oy)});
        }
        e->coord_bound0 = std::isfinite(cb) ? cb : 0.0;
        e->moves = 0;
    }
    // Road vertices (x, y, -F0, -(sigma+1)/2), padded with inert vertices.  Positions are offsets from the origin of
    // their tile of 1024 consecutive vertices (the centre of its box, rounded to 1/4 m): consecutive vertices of a
    // polyline are neighbours, so a vertex resolves to 2^-24 of ~50 m whatever the extent of the scene, and the road
    // kernel forms receiver - vertex relative to the tile (csf_pair.hip: road_kernel).
    d.nv = (int64_t)e->h_road.size() / 4;
    d.nv_pad = (d.nv + 63) / 64 * 64;
    if (d.nv > 0) {
        const int64_t tiles = (d.nv_pad + 1023) / 1024;
        if ((size_t)d.nv_pad > e->rv.n) HIPCHK(e, e->rv.alloc((size_t)d.nv_pad));
        if ((size_t)tiles > e->rvo.n) HIPCHK(e, e->rvo.alloc((size_t)tiles));
        std::vector<float4> rv((size_t)d.nv_pad, make_float4(1e15f, 1e15f, 0.f, -1.f));
        std::vector<float2> rvo((size_t)tiles, make_float2(0.f, 0.f));
        for (int64_t t = 0; t < tiles; t++) {
            const int64_t k0 = t * 1024, k1 = std::min<int64_t>(d.nv, k0 + 1024);
            double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
            for (int64_t k = k0; k < k1; k++) {
                x0 = std::min(x0, e->h_road[4 * k]), x1 = std::max(x1, e->h_road[4 * k]);
                y0 = std::min(y0, e->h_road[4 * k + 1]), y1 = std::max(y1, e->h_road[4 * k + 1]);
            }
            if (!(k1 > k0)) continue;
            const double tx = 0.25 * std::nearbyint(4.0 * (0.5 * (x0 + x1) - d.ox)), ty = 0.25 * std::nearbyint(4.0 * (0.5 * (y0 + y1) - d.oy));
            rvo[(size_t)t] = make_float2((float)tx, (float)ty);
            for (int64_t k = k0; k < k1; k++)
                rv[(size_t)k] = make_float4((float)((e->h_road[4 * k] - d.ox) - tx), (float)((e->h_road[4 * k + 1] - d.oy) - ty),
                                            (float)(-e->h_road[4 * k + 2]), (float)(-0.5 * (e->h_road[4 * k + 3] + 1.0)));
        }
        HIPCHK(e, hipMemcpy(e->rv.p, rv.data(), rv.size() * sizeof(float4), hipMemcpyHostToDevice));
        HIPCHK(e, hipMemcpy(e->rvo.p, rvo.data(), rvo.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
    d.rvo = e->rvo.p;
    d.road_np = 0;
    if (d.nv > 0) {   // one integer sigma for every edge: r^-(sigma+1) as a power of rsq(r^2)
        const double sg = e->h_road[3];
        bool same = sg == std::floor(sg) && sg >= 1 && sg <= 5;
        for (int64_t k = 1; same && k < d.nv; k++) same = e->h_road[4 * k + 3] == sg;
        if (same) d.road_np = (int32_t)sg + 1;
    }
    d.rv = e->rv.p;
    d.rg_nx = d.rg_ny = 0;
    if (road_grid) {
        const int rc = build_road_grid(e, rbox);
        if (rc) return rc;
    }
    set_shard(e);
    if ((size_t)d.n_pad > e->rec.n) return fail(e, CSF_E_CAPACITY, "record buffer too small for this shard layout");
    HIPCHK(e, hipMemsetAsync(e->part.p, 0, e->part.n * sizeof(float2), e->main));   // a new layout starts from clean partial sums
    // records as offsets from the scene origin first (what the sort keys are made of); the re-binning then gives every
    // record its own origin and rewrites it from the fp64 state just uploaded
    HIPCHK(e, hipMemsetAsync(e->rorg.p, 0, e->rorg.n * sizeof(float2), e->main));
    e->state_all_current = true;
    e->xbuf_fresh = false;                                       // (exchange records of the population as it was: every record is rewritten from the state)
    launch_records(d, e->main);
    int rrc = rebin(e);
    if (rrc) return rrc;
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(e->main));
    e->gather_pending = false;
    e->dirty = false;
    // what the side-by-side tick needs beside these arrays (enqueue_chase_tick): allocated here, where the host waits anyway - an
    // allocation synchronises, and the first such tick may lie inside a region somebody is timing
    if (chase_shape(e)) {
        int rcc = chase_alloc(e);
        if (rcc) return rcc;
    }
    return CSF_OK;
}

// A rank of a sharded run integrates only its own block of slots: its fp64 copy of the other blocks goes stale with the first
// tick.  Before the host may change the population - the road users of SUMO co-simulation arrive and leave on every rank's copy
// alike, intersection.py:458-634 - every rank needs every block's state as it is now: one all-gather per array on packed blocks
// (gather_population; a rare call); the members of a loopback group copy them from each other.  Every rank has to make the same population calls in the same order (they are collective from here on).
template <class F>
static int each_slot_array(csf_engine *e, F &&f) {                // f(base pointer, bytes per element, rows): the arrays download_all reads
    const int hl = e->d.hist_len;
    int rc;
#define ARR(buf, rows) if ((rc = f((char *)e->buf.p, sizeof(*e->buf.p), (int)(rows)))) return rc;
    ARR(s, STATE_ROWS) ARR(znp, 3) ARR(hx, hl) ARR(hy, hl) ARR(lti, 5) ARR(ppsi, 1) ARR(F, 6)
    ARR(ptr, 1) ARR(ti, 1) ARR(dgood, 1) ARR(znav, 1) ARR(zrid, 1) ARR(status, 1)
#undef ARR
    return CSF_OK;
}

static int gather_population(csf_engine *e) {
    // One exchange per ARRAY (thirteen in all), on equal blocks: a rank packs the rows of its own block - strided by the capacity -
    // into its place of a staging buffer ([rank][row][shard]: the in-place layout of ncclAllGather, the collective every tick
    // already uses), and unpacks the others' behind the call.  The members of a loopback group pack and unpack with the same code
    // and copy the packed blocks from each other where a real run calls the collective; a communicator of ONE rank takes the path
    // too (its all-gather is a copy onto itself) - so what the tests on one device cannot reach is the collective call alone.
    // (Until round 6: one ncclBroadcast per rank, array and ROW in one group - ~2 300 calls at 8 ranks with the ring's 128
    // columns - on a branch no test could enter.)
    const bool rehearse = e->nccl && e->world == 1 && !e->loopback;
    if (e->state_all_current && !rehearse) return CSF_OK;
    std::vector<csf_engine *> mem;
    if (e->loopback) mem.assign(e->group.begin(), e->group.end());
    else if (e->nccl) mem.push_back(e);
    if (mem.empty()) {                                             // (one device, no communicator: every block is its own)
        e->state_all_current = true;
        return CSF_OK;
    }
    const int world = e->loopback ? (int)mem.size() : e->world;
    const size_t cap = (size_t)e->cap;
    const int64_t shard = e->d.n_pad / world;
    if (shard <= 0 || shard * world != e->d.n_pad) return fail(e, CSF_E_STATE, "shard layout: %lld slots do not split over %d ranks", (long long)e->d.n_pad, world);
    for (csf_engine *m : mem) {
        if (m->cap != e->cap || m->d.n_pad != e->d.n_pad || m->d.n != e->d.n) return fail(e, CSF_E_STATE, "the ranks of a group hold different populations");
        HIPCHK(m, hipStreamSynchronize(m->main));
        if (m->comm) HIPCHK(m, hipStreamSynchronize(m->comm));
    }
    struct Arr { char *p; size_t el; int rows; };
    std::vector<std::vector<Arr>> arrs(mem.size());
    for (size_t i = 0; i < mem.size(); i++)
        each_slot_array(mem[i], [&](char *p, size_t el, int rows) -> int { arrs[i].push_back({p, el, rows}); return CSF_OK; });
    std::vector<DevBuf<char>> stage(mem.size());
    auto block = [&](int r, int64_t &lo, int64_t &hi) { lo = std::min<int64_t>(e->d.n, (int64_t)r * shard), hi = std::min<int64_t>(e->d.n, lo + shard); };
    int rc = CSF_OK;
    auto body = [&]() -> int {
        for (size_t k = 0; k < arrs[0].size(); k++) {
            const size_t el = arrs[0][k].el;
            const int rows = arrs[0][k].rows;
            if (rows <= 0) continue;
            const size_t blk = (size_t)rows * (size_t)shard * el;  // bytes of one rank's packed block
            int64_t lo, hi;
            for (size_t i = 0; i < mem.size(); i++) {              // pack: own block, rows made contiguous
                csf_engine *m = mem[i];
                HIPCHK(m, stage[i].reserve(blk * (size_t)world));
                block(m->rank, lo, hi);
                if (hi > lo)
                    HIPCHK(m, hipMemcpy2DAsync(stage[i].p + (size_t)m->rank * blk, (size_t)shard * el, arrs[i][k].p + (size_t)lo * el, cap * el,
                                               (size_t)(hi - lo) * el, (size_t)rows, hipMemcpyDeviceToDevice, m->main));
            }
            if (e->loopback) {                                     // the "all-gather" of a group on one device
                for (size_t i = 0; i < mem.size(); i++)
                    for (size_t j = 0; j < mem.size(); j++)
                        if (i != j)
                            HIPCHK(mem[i], hipMemcpyAsync(stage[i].p + (size_t)mem[j]->rank * blk, stage[j].p + (size_t)mem[j]->rank * blk, blk,
                                                          hipMemcpyDeviceToDevice, e->main));
            } else {
                NCCLCHK(e, g_rccl.AllGather(stage[0].p + (size_t)e->rank * blk, stage[0].p, blk, ncclChar, e->nccl, e->main));
            }
            for (size_t i = 0; i < mem.size(); i++) {              // unpack: everybody else's block
                csf_engine *m = mem[i];
                for (int r = 0; r < world; r++) {
                    if (r == m->rank) continue;
                    block(r, lo, hi);
                    if (hi > lo)
                        HIPCHK(m, hipMemcpy2DAsync(arrs[i][k].p + (size_t)lo * el, cap * el, stage[i].p + (size_t)r * blk, (size_t)shard * el,
                                                   (size_t)(hi - lo) * el, (size_t)rows, hipMemcpyDeviceToDevice, m->main));
                }
            }
        }
        return CSF_OK;
    };
    rc = body();
    hipError_t sr = hipSuccess;
    for (csf_engine *m : mem) {                                    // (before the staging buffers go)
        hipError_t r1 = hipStreamSynchronize(m->main);
        if (sr == hipSuccess) sr = r1;
    }
    for (auto &b : stage) b.release();
    if (rc) return rc;
    HIPCHK(e, sr);
    for (csf_engine *m : mem) m->state_all_current = true;
    return CSF_OK;
}

int prepare_mutation(csf_engine *e) {
    // a rank integrates only its own block: its fp64 copy of the other blocks goes stale with the first tick, and an
    // upload would rebuild their records from it - so the blocks are gathered first (gather_population)
    int rc;
    if ((e->world > 1 || e->loopback || e->nccl) && e->device_ahead) {
        if ((rc = flush_pending(e))) return rc;
        if ((rc = gather_population(e))) return rc;
    }
    rc = download_all(e);
    if (rc) return rc;
    e->dirty = true;
    return CSF_OK;
}

// Population changes between ticks go straight to the device arrays (spawn / retire / requeue kernels of csf_agent.hip)
// when the device copy is current: no download, no upload, no re-sort.  Sharded engines, engines with the opt-in
// history ring and CSF_INCREMENTAL=0 take the round trip through the host mirror (prepare_mutation / upload_all).
bool can_patch_device(const csf_engine *e) {
    return e->knobs.incremental && e->incremental && !e->dirty && e->world == 1 && !e->nccl && !e->loopback && e->d.hist == nullptr;
}

// a pinned, device-visible host buffer of `bytes` from the ring; waits only if the ring of four is exhausted
int stage_acquire(csf_engine *e, size_t bytes, csf_engine::PinnedSlot **out) {
    csf_engine::PinnedSlot &sl = e->pinned[e->pinned_next];
    e->pinned_next = (e->pinned_next + 1) % 4;
    if (sl.busy) {
        HIPCHK(e, hipEventSynchronize(sl.done));
        sl.busy = false;
    }
    if (bytes > sl.bytes) {
        if (sl.host) HIPCHK(e, hipHostFree(sl.host));
        sl.host = nullptr;
        sl.bytes = std::max<size_t>(2 * bytes, 1 << 16);
        HIPCHK(e, hipHostMalloc(&sl.host, sl.bytes, hipHostMallocMapped));
        HIPCHK(e, hipHostGetDevicePointer(&sl.dev, sl.host, 0));
    }
    if (!sl.done) HIPCHK(e, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    *out = &sl;
    return CSF_OK;
}

// The collected retirements, spawns and queue replacements -> one patch_kernel launch (csf_agent.hip), which also renews
// the circles of the tail batches the arrivals went to; the binned order is renewed when ticks x arrivals since the last
// re-binning says so (bounds_before_pair).
int flush_pending(csf_engine *e) {
    csf_engine::Pending &pd = e->pend;
    if (pd.empty()) return CSF_OK;
    Dev &d = e->d;
    e->mid_synced = false;
    const size_t b_ret = pd.retire.size() * sizeof(int32_t), b_sp = pd.spawn.size() * sizeof(SpawnRec),
                 b_rq = pd.requeue.size() * sizeof(QueueRec), b_rows = pd.rows.size() * sizeof(double);
    auto up8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
    PatchHeader h;
    h.n_retire = (int64_t)pd.retire.size();
    h.n_spawn = (int64_t)pd.spawn.size();
    h.n_requeue = (int64_t)pd.requeue.size();
    h.n_rows = (int64_t)(pd.rows.size() / 3);
    h.off_retire = (int64_t)up8(sizeof(PatchHeader));
    h.off_spawn = h.off_retire + (int64_t)up8(b_ret);
    h.off_requeue = h.off_spawn + (int64_t)up8(b_sp);
    h.off_rows = h.off_requeue + (int64_t)up8(b_rq);
    h.q_top = e->q_top;
    for (SpawnRec &r : pd.spawn) r.qbeg += e->q_top;          // relative to this batch's rows until now
    for (QueueRec &r : pd.requeue) r.qbeg += e->q_top;
    csf_engine::PinnedSlot *pin = nullptr;
    int rc = stage_acquire(e, (size_t)h.off_rows + b_rows, &pin);
    if (rc) return rc;
    char *base = (char *)pin->host;
    memcpy(base, &h, sizeof h);
    if (b_ret) memcpy(base + h.off_retire, pd.retire.data(), b_ret);
    if (b_sp) memcpy(base + h.off_spawn, pd.spawn.data(), b_sp);
    if (b_rq) memcpy(base + h.off_requeue, pd.requeue.data(), b_rq);
    if (b_rows) memcpy(base + h.off_rows, pd.rows.data(), b_rows);
    // The arrivals went to known places of the sentinel tail (rebin): the launch itself renews the circles of those few
    // batches, and the circles the last pair launch emitted for this tick stay good for all others (a retirement only
    // leaves a circle larger than necessary).  Otherwise every circle is recomputed before the next pair launch.
    int b0 = 0, b1 = 0;
    bool circles_here = e->bounds_fresh && e->tail_tracked && d.classify && !e->pend_inplace;
    if (circles_here && h.n_spawn > 0) {
        b0 = (int)((e->live_at_rebin + e->tail_flushed) / 64);
        b1 = (int)((e->live_at_rebin + e->tail_used + 63) / 64);
        if (b1 - b0 > 32 || (int64_t)b1 * 64 > d.n_src) circles_here = false, b0 = b1 = 0;
    }
    launch_patch(d, h, pin->dev, e->ticket.p, b0, b1, h.n_retire + h.n_spawn + h.n_requeue + 3 * h.n_rows, e->main);
    e->tail_flushed = e->tail_used;
    e->pend_inplace = false;
    // The class-segmented order has no tail for arrivals: an arrival belongs into its set's run, which only a re-binning
    // can give it (the sort reads every slot's record and set) - so the order is renewed before the next pair launch
    // (~50 us, against 5 ms for the way through the host mirror).  Departures leave sentinel records in their runs.
    if (!e->segs.empty() && h.n_spawn > 0) e->ticks_since_rebin = std::max<int64_t>(e->ticks_since_rebin, e->knobs.rebin_ticks);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipEventRecord(pin->done, e->main));
    pin->busy = true;
    e->q_top += h.n_rows;
    for (int32_t a : pd.retire) e->sidx[(size_t)a].retire = -1, e->dev_alive[(size_t)a] = 0;
    for (const SpawnRec &r : pd.spawn) e->sidx[(size_t)r.slot].spawn = -1, e->dev_alive[(size_t)r.slot] = 1;
    for (const QueueRec &r : pd.requeue) e->sidx[(size_t)r.slot].requeue = -1;
    e->churn += e->pend_tail_spawns;                          // (an arrival in a leaver's slot adds nothing to the tail)
    e->pend_tail_spawns = 0;
    pd.retire.clear();
    pd.spawn.clear();
    pd.requeue.clear();
    pd.rows.clear();
    if (!circles_here) e->bounds_fresh = false;
    e->device_ahead = true;                                    // the host mirror of the patched slots was not kept up
    d.n_live = (int64_t)e->order.size();
    if (e->knobs.fake_world <= 1) d.hi = d.n;                 // (unsharded: the receiver block is every slot)
    update_far_radius(e);
    return CSF_OK;
}

// the queue replacement collected for slot a, if any, is void (the slot is retired or spawned into anew)
void drop_pending_requeue(csf_engine *e, size_t a) {
    const int32_t at = e->sidx[a].requeue;
    if (at < 0) return;
    const QueueRec last = e->pend.requeue.back();
    e->pend.requeue[(size_t)at] = last;
    e->sidx[(size_t)last.slot].requeue = at;
    e->pend.requeue.pop_back();
    e->sidx[a].requeue = -1;
}

int sync_order(csf_engine *e) {                         // the device copy of the population order (read-back kernels)
    Dev &d = e->d;
    bool identity = (int64_t)e->order.size() == d.n;
    for (size_t i = 0; identity && i < e->order.size(); i++) identity = e->order[i] == (int32_t)i;
    if (identity) {
        d.order = nullptr;
    } else {
        if (e->order_dirty)
            HIPCHK(e, hipMemcpyAsync(e->order_dev.p, e->order.data(), e->order.size() * sizeof(int32_t), hipMemcpyHostToDevice, e->main));
        d.order = e->order_dev.p;
    }
    e->order_dirty = false;
    return CSF_OK;
}

// slot-indexed device array (component c at [c * cap + slot]) -> caller's array in population order
template <class T>
int read_rows(csf_engine *e, const T *dev, int comps, T *out, bool row_major) {
    const int64_t n = (int64_t)e->order.size(), ns = e->d.n, cap = e->cap;
    std::vector<T> tmp((size_t)ns);
    for (int c = 0; c < comps; c++) {
        HIPCHK(e, hipMemcpy(tmp.data(), dev + (size_t)c * cap, (size_t)ns * sizeof(T), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; i++) out[row_major ? i * comps + c : c * n + i] = tmp[(size_t)e->order[(size_t)i]];
    }
    return CSF_OK;
}

// entry points that index the device arrays by road user (replay, history, sharding) want slots == population order
int ensure_compact(csf_engine *e) {
    bool identity = e->free_tail.empty() && e->free_recent.empty() && (int64_t)e->order.size() == e->d.n;
    for (size_t i = 0; identity && i < e->order.size(); i++) identity = e->order[i] == (int32_t)i;
    if (identity) return CSF_OK;
    int rc = download_all(e);
    if (rc) return rc;
    e->dirty = true;
    return upload_all(e);
}

// CSF_COMM_STREAM=second puts the all-gather on a second HIP stream, so that the destination-force phase of the
// next tick (launched before the wait) overlaps it.  Measured with a 1-rank communicator the two cross-stream
// event waits per tick and the extra launch cost 26 us against 8 us in stream order, more than the ~6 us of
// destination-force work they can hide, so the default keeps the collective in stream order on the main stream.
bool comm_second_stream(const csf_engine *e) { return e->comm_second; }

int all_gather_records(csf_engine *e) {
    Dev &d = e->d;
    size_t shard = (size_t)(d.n_pad / e->world);
    const bool second = comm_second_stream(e);
    hipStream_t cs = second ? e->comm : e->main;
    if (second) HIPCHK(e, hipStreamWaitEvent(e->comm, e->ev_integ, 0));
    // ONE collective on the exchange records (csf_dev.h: xbuf; 32 B per slot: the record, what its position left over in
    // fp32, the Bicycle field's second record) - until round 5 a group of two or three all-gathers on rec / reclo / rec2
    if (d.xbuf == nullptr) return fail(e, CSF_E_DEVICE, "no exchange buffer (allocation failed when the shard layout was set)");
    NCCLCHK(e, g_rccl.AllGather(d.xbuf + 2 * (size_t)e->rank * shard, d.xbuf, shard * 8, ncclFloat32, e->nccl, cs));
    e->xbuf_fresh = true;
    if (second) {
        HIPCHK(e, hipEventRecord(e->ev_gather, e->comm));
        e->gather_pending = true;
    }
    return CSF_OK;
}

constexpr size_t PROF_SLOTS = 256, PROF_KEEP = 1 << 16;

// oldest outstanding slot -> running sums (waits for its last event)
int prof_resolve_one(csf_engine *e) {
    csf_engine::ProfSlot &sl = e->prof_pool[e->prof_resolved % PROF_SLOTS];
    const int last = sl.gather ? 7 : sl.agent ? 5 : sl.road ? 3 : sl.pair ? 1 : -1;
    if (last >= 0) HIPCHK(e, hipEventSynchronize(sl.ev[last]));
    const bool have[4] = {sl.pair, sl.road, sl.agent, sl.gather};
    for (int k = 0; k < 4; k++) {
        if (!have[k]) continue;
        float ms = 0;
        HIPCHK(e, hipEventElapsedTime(&ms, sl.ev[2 * k], sl.ev[2 * k + 1]));
        e->prof_ms[k] += ms;
        e->prof_cnt[k]++;
        if (e->prof_us[k].size() < PROF_KEEP) e->prof_us[k].push_back(ms * 1e3f);
    }
    // CSF_CHASE_CLOCK (measurement aid, every tick sampled): the dispatch time stamps of consecutive ticks against each other - pair end ->
    // per-agent start / end, per-agent end -> the next tick's pair start (the previous slot of the pool is still intact)
    if (!e->knobs.chase_clock.empty() && sl.pair && sl.agent && e->prof_resolved > 0) {
        csf_engine::ProfSlot &pv = e->prof_pool[(e->prof_resolved - 1) % PROF_SLOTS];
        float a = 0, b = 0, c = 0, d2 = 0;
        if (pv.pair && pv.agent && hipEventElapsedTime(&a, pv.ev[1], pv.ev[4]) == hipSuccess && hipEventElapsedTime(&b, pv.ev[1], pv.ev[5]) == hipSuccess &&
            hipEventElapsedTime(&c, pv.ev[5], sl.ev[0]) == hipSuccess && hipEventElapsedTime(&d2, pv.ev[0], sl.ev[0]) == hipSuccess)
            fprintf(stderr, "CHASE_EVENTS agent_start_after_pair_end %.2f agent_end_after_pair_end %.2f next_pair_start_after_agent_end %.2f period %.2f\n",
                    a * 1e3, b * 1e3, c * 1e3, d2 * 1e3);
        (void)hipGetLastError();
    }
    e->prof_resolved++;
    return CSF_OK;
}

int prof_make_pool(csf_engine *e) {
    if (!e->prof_pool.empty()) return CSF_OK;
    e->prof_pool.resize(PROF_SLOTS);
    for (auto &sl : e->prof_pool)
        for (auto &ev : sl.ev) HIPCHK(e, hipEventCreate(&ev));
    return CSF_OK;
}

// a free slot for this tick (NULL with *rc == 0: profiling is off for it)
csf_engine::ProfSlot *prof_slot(csf_engine *e, int *rc) {
    *rc = CSF_OK;
    if (e->profile <= 0 || e->d.tick % e->profile != 0) return nullptr;
    if ((*rc = prof_make_pool(e))) return nullptr;
    if (e->prof_issued - e->prof_resolved >= PROF_SLOTS && (*rc = prof_resolve_one(e))) return nullptr;
    csf_engine::ProfSlot *sl = &e->prof_pool[e->prof_issued % PROF_SLOTS];
    sl->pair = sl->road = sl->agent = sl->gather = false;
    e->prof_issued++;
    return sl;
}

int wait_gather(csf_engine *e) {
    if (e->gather_pending) {
        HIPCHK(e, hipStreamWaitEvent(e->main, e->ev_gather, 0));
        e->gather_pending = false;
    }
    return CSF_OK;
}

}  // namespace

// ================================================================================ C ABI ======

extern "C" {

int32_t csf_abi_version(void) { return CSF_ABI_VERSION; }

const char *csf_last_error(const csf_engine *e) { return e ? e->err.c_str() : g_create_error.c_str(); }

csf_engine *csf_create(const csf_params *params, int64_t n_capacity, int32_t device) {
    if (check_params(nullptr, params)) return nullptr;
    if (n_capacity < 1) {
        fail(nullptr, CSF_E_ARG, "n_capacity must be >= 1");
        return nullptr;
    }
    int ndev = 0;
    hipError_t r = hipGetDeviceCount(&ndev);
    if (r != hipSuccess || ndev <= 0) {
        fail(nullptr, CSF_E_DEVICE, "no HIP device available (%s); this engine has no CPU fallback",
             r != hipSuccess ? hipGetErrorString(r) : "device count 0");
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        fail(nullptr, CSF_E_DEVICE, "device %d out of range (%d visible)", device, ndev);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        fail(nullptr, CSF_E_DEVICE, "hipSetDevice(%d) failed", device);
        return nullptr;
    }
    if (const char *pv = getenv("CSF_DEBUG_POISON")) g_poison = atoi(pv) != 0;   // (before the first allocation)
    csf_engine *e = new csf_engine();
    e->device = device;
    e->cap_user = n_capacity;
    // a binned population keeps up to 4096 sentinel slots behind its real batches for arrivals (rebin(), csf_add_agents)
    e->cap = n_capacity + (n_capacity >= BIN_MIN_AGENTS ? std::min<int64_t>(TAIL_SLOTS, n_capacity / 4) : 0);
    // whole batches of 64 slots: n_pad (set_shard) is rounded up to a multiple of 64 and must stay <= cap, the stride of
    // every SoA array - fresh slots are handed out up to n_pad (csf_add_agents)
    e->cap = (e->cap + 63) / 64 * 64;
    e->knobs.read();
    e->time_pop = getenv("CSF_TIME_POP") != nullptr;
    e->d.p = *params;
    derive_consts(e);
    auto bail = [&](const char *what) {
        g_create_error = std::string(what) + ": " + e->err;
        csf_destroy(e);
        return (csf_engine *)nullptr;
    };
    if (hipStreamCreateWithFlags(&e->main, hipStreamNonBlocking) != hipSuccess) return bail("stream");
    e->main_hold = std::make_shared<csf_engine::StreamHold>();
    e->main_hold->s = e->main;
    {   // The second stream from ANOTHER priority level: the runtime hands streams of one level out over a small pool of hardware
        // queues (four by default), and two streams that share a queue serialise - the per-agent kernel beside the pair launch
        // (enqueue_chase_tick) then runs behind it after all, with a third launch on top (measured: 120 against 115 us per tick with
        // two engines in one process; 111 with queues of their own).  Priority levels have pools of their own.
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);           // (numerically lower = higher priority)
        if (hi < lo && hipStreamCreateWithPriority(&e->comm, hipStreamNonBlocking, hi) != hipSuccess) e->comm = nullptr;
        (void)hipGetLastError();
        if (!e->comm && hipStreamCreateWithFlags(&e->comm, hipStreamNonBlocking) != hipSuccess) return bail("stream");
    }
    if (hipEventCreateWithFlags(&e->ev_integ, hipEventDisableTiming) != hipSuccess) return bail("event");
    if (hipEventCreateWithFlags(&e->ev_gather, hipEventDisableTiming) != hipSuccess) return bail("event");
    if (alloc_all(e) != CSF_OK) return bail("allocation");
    set_shard(e);
    return e;
}

size_t csf_params_size(void) { return sizeof(csf_params); }

csf_engine *csf_create_v(const csf_params *params, size_t params_size, int32_t abi_version, int64_t n_capacity, int32_t device) {
    // before anything is read from `params`: a hand-declared struct of another ABI is shorter (or longer) than ours
    if (params_size != sizeof(csf_params) || abi_version != CSF_ABI_VERSION) {
        fail(nullptr, CSF_E_ABI, "csf_create_v: the caller's csf_params has %zu bytes and ABI %d, this library's has %zu bytes and ABI %d "
             "(include/csf.h: field order and size are ABI)", params_size, (int)abi_version, sizeof(csf_params), (int)CSF_ABI_VERSION);
        return nullptr;
    }
    return csf_create(params, n_capacity, device);
}

int csf_destroy(csf_engine *e) {
    if (!e) return CSF_OK;
    (void)hipSetDevice(e->device);
    if (e->main) (void)hipStreamSynchronize(e->main);
    if (e->comm) (void)hipStreamSynchronize(e->comm);
    if (e->time_pop && e->tp_calls > 0)
        fprintf(stderr, "CSF_TIME_POP per call (us): remove: loop %.1f (of it holes_add %.1f) compaction %.1f | add: slots %.1f (of it holes_take %.1f) fill %.1f | queues %.1f  [%lld calls]\n",
                e->tp_ns[0] / 1e3 / e->tp_calls, e->tp_ns[1] / 1e3 / e->tp_calls, e->tp_ns[2] / 1e3 / e->tp_calls, e->tp_ns[3] / 1e3 / e->tp_calls,
                e->tp_ns[4] / 1e3 / e->tp_calls, e->tp_ns[5] / 1e3 / e->tp_calls, e->tp_ns[6] / 1e3 / e->tp_calls, (long long)e->tp_calls);
    if (e->chase_clock.p) {
        std::vector<unsigned long long> h(128 * 8);
        if (hipMemcpy(h.data(), e->chase_clock.p, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost) == hipSuccess) {
            static std::atomic<int> g_clock_files{0};     // (one file per engine of the process, in the order they are destroyed)
            const int fk = g_clock_files.fetch_add(1);
            const std::string path = fk == 0 ? e->knobs.chase_clock : e->knobs.chase_clock + "." + std::to_string(fk);
            if (FILE *f = fopen(path.c_str(), "wb")) {
                fwrite(h.data(), sizeof(h[0]), h.size(), f);
                fwrite(&e->chase_ticks, sizeof e->chase_ticks, 1, f);
                fclose(f);
            }
        }
        e->chase_clock.release();
    }
    if (e->atrace.p) {  // CSF_TRACE_AGENT=<file>: the stamps of the last per-agent launch
        std::vector<uint64_t> h(e->atrace.n);
        if (hipMemcpy(h.data(), e->atrace.p, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE *f = fopen(e->knobs.trace_agent.c_str(), "wb")) {
                fwrite(h.data(), sizeof(uint64_t), h.size(), f);
                fclose(f);
            }
        }
        e->atrace.release();
    }
    if (e->trace.p) {  // CSF_TRACE_BLOCKS=<file>: workgroup timeline of the last pair-kernel launch
        std::vector<uint64_t> h(e->trace_words);
        if (hipMemcpy(h.data(), e->trace.p, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE *f = fopen(e->knobs.trace_blocks.c_str(), "wb")) {
                fwrite(h.data(), sizeof(uint64_t), h.size(), f);
                fclose(f);
            }
        }
        e->trace.release();
    }
    if (e->nccl && g_rccl.CommDestroy) g_rccl.CommDestroy(e->nccl);
    for (auto &sl : e->prof_pool)
        for (hipEvent_t ev : sl.ev)
            if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->cal_ev)
        if (ev) (void)hipEventDestroy(ev);
    if (e->ev_integ) (void)hipEventDestroy(e->ev_integ);
    if (e->ev_gather) (void)hipEventDestroy(e->ev_gather);
    e->s.release(); e->vdes.release(); e->q.release(); e->znp.release(); e->hx.release(); e->hy.release();
    e->rec_alt.release(); e->recg_alt.release(); e->rec2_alt.release(); e->src64_a.release(); e->src64_b.release();
    e->recs_alt.release(); e->chase_cnt.release(); e->chase_misc.release();
    e->lti.release(); e->ppsi.release(); e->script.release(); e->sbeg.release(); e->slen.release(); e->F.release(); e->hist.release(); e->qbeg.release(); e->qlen.release(); e->alive.release(); e->order_dev.release();
    e->ptr.release(); e->ti.release(); e->dgood.release(); e->znav.release(); e->zrid.release();
    e->status.release(); e->rec.release(); e->rv.release(); e->rvo.release(); e->rg_v.release(); e->rg_start.release(); e->rg_c.release(); e->kat4.release(); e->rec2.release(); e->recs2.release();
    e->part.release(); e->froad.release(); e->kat2.release(); e->bnd.release(); e->bnd2.release(); e->rorg.release(); e->reclo.release(); e->xbuf.release(); e->segtab.release(); e->tcirc.release(); e->clist.release(); e->ccount.release(); e->far_stat.release(); e->edge.release(); e->edge_n.release(); e->edge_head.release(); e->perm.release(); e->pos.release(); e->recs.release(); e->recg.release(); e->recb.release(); e->borg.release();
    e->ticket.release(); e->scratch_u8.release(); e->scratch_i32.release(); e->scratch_f64.release(); e->scratch_cnt.release();
    e->ptab.release(); e->pctab.release(); e->pbtab.release(); e->cls.release();
    e->sort_vals.release(); e->rlist.release(); e->sort_keys.release(); e->sort_keys_out.release(); e->sort_tmp.release();
    if (e->snap_host) (void)hipHostFree(e->snap_host);
    if (e->bound_pin) (void)hipHostFree(e->bound_pin);
    if (e->holes.pos) (void)hipHostFree(e->holes.pos);
    if (e->holes.bnd) (void)hipHostFree(e->holes.bnd);
    if (e->holes.ev) (void)hipEventDestroy(e->holes.ev);
    for (auto &sl : e->pinned) {
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (e->comm) (void)hipStreamDestroy(e->comm);
    for (csf_engine *m : e->group)          // the others of a loopback group lose this member: no group any more (csf_step_group says so)
        if (m != e) m->group.clear();
    e->main = nullptr;
    e->main_hold.reset();                   // (the stream itself goes with its last holder; this engine's events went above)
    delete e;
    return CSF_OK;
}

int64_t csf_num_agents(const csf_engine *e) { return e ? (int64_t)e->order.size() : 0; }
int32_t csf_num_states(const csf_engine *e) { return e ? e->d.ns : 0; }

// What a road user's constructor makes of vehicle.s beside it, with the limits of ITS parameter set: the state of the rider
// model's integrator and the riding / walking state (vehicle.py:1728-1736; dynamics.py:195-197, 828, 987-993).  For a
// BalancingRiderBicycle (dynamics.py:306-307, 350-371): roll, steer, their rates and the yaw in the bike model's frame
// (steer, yaw mirrored), and the speed its first gains belong to.
static void side_state(csf_engine *e, size_t a, const csf_params &p) {
    const size_t cap = (size_t)e->cap;
    const double *s = e->h_s.data() + a;
    const double delta = s[4 * cap];
    e->h_zrid[a] = s[3 * cap] < p.v_max_walk ? 0 : 1;            // vehicle.py:1732-1736
    e->h_dgood[a] = (-p.delta_max_walk < delta && p.delta_max_walk > delta) ? 1 : 0;
    if (p.model == CSF_BALANCINGRIDER) {
        e->h_lti[0 * cap + a] = s[5 * cap];
        e->h_lti[1 * cap + a] = -s[4 * cap];
        e->h_lti[2 * cap + a] = s[7 * cap];
        e->h_lti[3 * cap + a] = -s[6 * cap];
        e->h_lti[4 * cap + a] = -s[2 * cap];
        e->h_ppsi[a] = s[3 * cap];
        return;
    }
    e->h_lti[0 * cap + a] = delta;                               // vehicle.py:1728
    e->h_lti[1 * cap + a] = 0.0;
    e->h_lti[2 * cap + a] = s[5 * cap];
    e->h_lti[3 * cap + a] = 0.0;
    e->h_lti[4 * cap + a] = s[2 * cap];
    e->h_ppsi[a] = s[2 * cap];                                   // dynamics.py:828, 987-993
}

// q_off / q_rows: NULL (every new road user gets the one-row queue of vehicle.py:183-185) or the destination queues the arrivals
// START with, CSR - what csf_add_agents followed by csf_set_dest_queue(reset = 1) on the new road users leaves, in one pass
// over them (csf_replace_agents)
static int add_agents_impl(csf_engine *e, int64_t n, const double *s0, const double *v_desired, const int64_t *q_off, const double *q_rows) {
    if (!e) return CSF_E_ARG;
    if (n < 0 || (n > 0 && (!s0 || !v_desired))) return fail(e, CSF_E_ARG, "csf_add_agents: bad arguments");
    int64_t q_total = 0;
    if (q_off != nullptr) {
        if (!q_rows) return fail(e, CSF_E_ARG, "csf_replace_agents: queue offsets without rows");
        for (int64_t k = 0; k < n; k++) {
            if (q_off[k + 1] <= q_off[k]) return fail(e, CSF_E_ARG, "csf_replace_agents: an arrival's destination queue must have a row");
        }
        q_total = n > 0 ? q_off[n] - q_off[0] : 0;
    }
    if ((int64_t)e->order.size() + n > e->cap_user) return fail(e, CSF_E_CAPACITY, "capacity %lld exceeded", (long long)e->cap_user);
    if (n == 0) return CSF_OK;
    HIPCHK(e, hipSetDevice(e->device));
    Dev &d = e->d;
    const int64_t reuse = std::min<int64_t>(n, (int64_t)(e->free_tail.size() + e->free_recent.size()));
    // collected for the device when its copy is current, the new slots have a place in the binned order (slot < n_pad)
    // and the start rows fit behind the queues already in the slab
    bool patch = can_patch_device(e) && d.n + (n - reuse) <= d.n_pad;
    if (patch && e->q_top + (int64_t)(e->pend.rows.size() / 3) + std::max(n, q_total) > d.qcap) {
        int rc = compact_slab(e, std::max(n, q_total));
        if (rc) return rc;
    }
    if (!patch) {
        int rc = prepare_mutation(e);
        if (rc) return rc;
    }
    const int ns = d.ns;
    const int64_t cap = e->cap;
    const csf_params &p = d.p;
    // first the slots (an arrival in a leaver's slot lands anywhere in the per-slot arrays of the host: asked for ahead of the
    // loop that fills them - the loop's time was cache misses, 55 us per tick with 819 arrivals), then the road users
    std::vector<int32_t> &slot_of = e->add_slots;
    slot_of.resize((size_t)n);
    const auto ta0 = std::chrono::steady_clock::now();
    int64_t ta_holes = 0;
    for (int64_t k = 0; k < n; k++) {
        int64_t a;
        bool tail = true;
        int32_t hole = -1;
        if (patch && e->tail_tracked && e->knobs.hole_reuse && !e->free_recent.empty() && holes_ready(e)) {
            if (e->time_pop) {
                const auto h0 = std::chrono::steady_clock::now();
                hole = holes_take(e, s0[k * ns], s0[k * ns + 1]);
                ta_holes += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - h0).count();
            } else {
                hole = holes_take(e, s0[k * ns], s0[k * ns + 1]);      // the slot of a road user that left from around here
            }
        }
        if (hole >= 0) {
            a = hole;
            tail = false;
            e->pend_inplace = true;                              // (the circles are renewed before the next pair launch)
            d.clist = nullptr;
        } else if (!e->free_tail.empty()) {                      // the next place of the sentinel tail
            a = e->free_tail.back();
            e->free_tail.pop_back();
        } else if (patch ? d.n < d.n_pad : e->free_recent.empty()) {   // a fresh slot: the tail continues there
            a = d.n++;
        } else if (!e->free_recent.empty()) {                    // (inside a real batch: only when nothing else is left)
            a = e->free_recent.back();
            e->free_recent.pop_back();
            e->sidx[(size_t)a].recent = -1;
            tail = false;
            e->pend_inplace = true;
            d.clist = nullptr;                                   // (a real batch's circle stretches: no candidate lists until the re-binning)
        } else {
            a = d.n++;
        }
        if (tail) e->tail_used++, e->pend_tail_spawns += patch ? 1 : 0;
        slot_of[(size_t)k] = (int32_t)a;
        __builtin_prefetch(&e->h_vdes[(size_t)a], 1);
        __builtin_prefetch(&e->h_q[(size_t)a], 1);
        __builtin_prefetch(&e->h_alive[(size_t)a], 1);
        __builtin_prefetch(&e->h_cls[(size_t)a], 1);
        __builtin_prefetch(&e->sidx[(size_t)a], 1);
    }
    const auto ta1 = std::chrono::steady_clock::now();
    for (int64_t k = 0; k < n; k++) {
        const int64_t a = slot_of[(size_t)k];
        const double *s = s0 + k * ns;
        e->h_vdes[a] = v_desired[k];
        const double *qr = q_off ? q_rows + 3 * q_off[k] : nullptr;         // its queue: given, or the start row (vehicle.py:183-185)
        const int32_t qn = q_off ? (int32_t)(q_off[k + 1] - q_off[k]) : 1;
        if (qr) e->h_q[a].assign(qr, qr + 3 * (size_t)qn);
        else e->h_q[a].assign({s[0], s[1], 0.0});
        e->h_alive[a] = 1;
        e->h_cls[a] = 0;
        e->order.push_back((int32_t)a);
        if (patch) {                                             // a record for the patch kernel, which writes the rest
            if (e->sidx[(size_t)a].retire >= 0) {             // the slot was freed in this batch: the spawn resets all of it
                const int32_t at = e->sidx[(size_t)a].retire, last = e->pend.retire.back();
                e->pend.retire[(size_t)at] = last;
                e->sidx[(size_t)last].retire = at;
                e->pend.retire.pop_back();
                e->sidx[(size_t)a].retire = -1;
            }
            drop_pending_requeue(e, (size_t)a);                  // (defensive: a retirement has dropped it already)
            SpawnRec r;
            r.slot = (int32_t)a;
            r.qlen = qn;
            r.qbeg = (int64_t)(e->pend.rows.size() / 3);
            for (int c = 0; c < STATE_ROWS; c++) r.s[c] = c < ns ? s[c] : 0.0;
            e->coord_bound0 = std::max({e->coord_bound0, std::fabs(s[0] - d.ox), std::fabs(s[1] - d.oy)});   // (set_fov_band)
            r.vdes = v_desired[k];
            r.cls = 0;
            r.pad = 0;
            if (qr) e->pend.rows.insert(e->pend.rows.end(), qr, qr + 3 * (size_t)qn);
            else e->pend.rows.insert(e->pend.rows.end(), {s[0], s[1], 0.0});
            e->sidx[(size_t)a].spawn = (int32_t)e->pend.spawn.size();
            e->pend.spawn.push_back(r);
            continue;
        }
        for (int c = 0; c < STATE_ROWS; c++) e->h_s[c * cap + a] = c < ns ? s[c] : 0.0;
        e->h_s[2 * cap + a] = limit_angle_h(s[2]);               // vehicle.py:154-155
        e->h_ptr[a] = 0;
        e->h_znav[a] = 0;                                        // vehicle.py:188
        for (int c = 0; c < 3; c++) e->h_znp[c * cap + a] = 0.0;
        e->h_ti[a] = 0;                                          // vehicle.py:146
        e->h_hx[a] = s[0];                                       // traj[:, 0] = s  (vehicle.py:159-160)
        e->h_hy[a] = s[1];
        side_state(e, (size_t)a, p);
        for (int c = 0; c < 6; c++) e->h_F[c * cap + a] = 0.0;
        e->h_status[a] = 0;
    }
    if (e->time_pop) {
        e->tp_ns[3] += std::chrono::duration_cast<std::chrono::nanoseconds>(ta1 - ta0).count();
        e->tp_ns[4] += ta_holes;
        e->tp_ns[5] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - ta1).count();
    }
    d.n_live = (int64_t)e->order.size();
    e->order_dirty = true;
    // The candidate tile lists were built for the receiver groups and the tiles of the last re-binning: an arrival sits in the
    // sentinel tail (or a fresh slot behind it) - its group had an empty circle then and lists nothing, and groups behind
    // the old last one have no list at all.  Until the next re-binning every group walks every tile again.
    d.clist = nullptr;
    d.ccount = nullptr;
    if (!patch) {
        set_shard(e);
    } else {
        // (without the binned order a slot is its own place: the sources end with the last slot in use)
        const int64_t n_src = e->tail_tracked ? std::min(d.n_pad, (e->live_at_rebin + e->tail_used + 63) / 64 * 64)
                              : (d.classify ? d.n_pad : std::min(d.n_pad, (d.n + 63) / 64 * 64));
        if (e->segs.empty() && (n_src > d.n_src || !e->tail_tracked)) {     // (segments: the chunks are the runs', until the re-binning)
            d.n_src = std::max(n_src, d.n_src);
            set_chunks(e);
        }
    }
    return CSF_OK;
}

int csf_add_agents(csf_engine *e, int64_t n, const double *s0, const double *v_desired) { return add_agents_impl(e, n, s0, v_desired, nullptr, nullptr); }

int csf_remove_agents(csf_engine *e, int64_t n, const int32_t *idx);

int csf_replace_agents(csf_engine *e, int64_t n_leave, const int32_t *idx_leave, int64_t n_arrive, const double *s0, const double *v_desired,
                       const int64_t *q_offsets, const double *q_rows) {
    if (!e) return CSF_E_ARG;
    if (n_leave < 0 || n_arrive < 0 || (n_arrive > 0 && (!q_offsets || !q_rows))) return fail(e, CSF_E_ARG, "csf_replace_agents: bad arguments");
    if ((int64_t)e->order.size() - n_leave + n_arrive > e->cap_user) return fail(e, CSF_E_CAPACITY, "capacity %lld exceeded", (long long)e->cap_user);
    int rc = n_leave > 0 ? csf_remove_agents(e, n_leave, idx_leave) : CSF_OK;
    if (rc) return rc;
    return n_arrive > 0 ? add_agents_impl(e, n_arrive, s0, v_desired, q_offsets, q_rows) : CSF_OK;
}

int csf_remove_agents(csf_engine *e, int64_t n, const int32_t *idx) {
    if (!e) return CSF_E_ARG;
    if (n < 0 || (n > 0 && !idx)) return fail(e, CSF_E_ARG, "csf_remove_agents: bad arguments");
    HIPCHK(e, hipSetDevice(e->device));
    const int64_t pop = (int64_t)e->order.size();
    // The listed indices in ascending order without repeats (what a caller that walks its population produces: the mirror's
    // remove_road_users_by_id, tools/churn_rate.py): no mark array over the population, and the survivors are moved down block by
    // block - with 5 % of 16 384 road users leaving per tick the mark-and-copy loop over all of them was 40 us of every tick.
    bool ascending = true;
    for (int64_t k = 0; k < n; k++) {
        if (idx[k] < 0 || idx[k] >= pop) return fail(e, CSF_E_ARG, "agent index %d out of range", idx[k]);
        if (k > 0 && idx[k] <= idx[k - 1]) ascending = false;
    }
    if (n == 0) return CSF_OK;
    const int32_t *lst = idx;
    std::vector<int32_t> &sorted = e->remove_sorted;
    if (!ascending) {                                            // any order, repeats allowed: sorted and made unique here
        sorted.assign(idx, idx + n);
        std::sort(sorted.begin(), sorted.end());
        sorted.erase(std::unique(sorted.begin(), sorted.end()), sorted.end());
        lst = sorted.data();
        n = (int64_t)sorted.size();
    }
    const bool patch = can_patch_device(e);
    if (!patch) {
        int rc = prepare_mutation(e);
        if (rc) return rc;
    }
    const auto tp0 = std::chrono::steady_clock::now();
    int64_t tp_holes = 0;
    for (int64_t k = 0; k < n; k++) {
        const int32_t a = e->order[(size_t)lst[k]];
        e->h_alive[(size_t)a] = 0;
        e->h_cls[(size_t)a] = 0;                                // (a dead slot's sentinel record is looked up in set 0: the table may shrink)
        e->h_q[(size_t)a].clear();
        e->h_script[(size_t)a].clear();
        e->sidx[(size_t)a].recent = (int32_t)e->free_recent.size();
        e->free_recent.push_back(a);
        if (!patch) continue;
        if (e->tail_tracked && e->knobs.hole_reuse) {
            if (e->time_pop) {
                const auto h0 = std::chrono::steady_clock::now();
                holes_add(e, a);
                tp_holes += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - h0).count();
            } else {
                holes_add(e, a);
            }
        }
        drop_pending_requeue(e, (size_t)a);                      // a queue collected for the road user that leaves
        if (e->sidx[(size_t)a].spawn >= 0) {                  // added and removed within one batch: never reaches the device
            const int32_t at = e->sidx[(size_t)a].spawn;
            const SpawnRec last = e->pend.spawn.back();
            e->pend.spawn[(size_t)at] = last;
            e->sidx[(size_t)last.slot].spawn = at;
            e->pend.spawn.pop_back();
            e->sidx[(size_t)a].spawn = -1;
            if (!e->dev_alive[(size_t)a]) continue;              // dead on the device, or never used: nothing to undo
        }                                                        // (else: its previous occupant is still alive there)
        e->sidx[(size_t)a].retire = (int32_t)e->pend.retire.size();
        e->pend.retire.push_back(a);
    }
    const auto tp1 = std::chrono::steady_clock::now();
    {   // the remaining road users keep their relative order: the blocks between two leavers move down
        int32_t *o = e->order.data();
        int64_t w = lst[0];
        for (int64_t k = 0; k < n; k++) {
            const int64_t from = (int64_t)lst[k] + 1, to = k + 1 < n ? (int64_t)lst[k + 1] : pop;
            if (to > from) std::memmove(o + w, o + from, (size_t)(to - from) * sizeof(int32_t));
            w += to - from;
        }
        e->order.resize((size_t)w);
    }
    if (e->time_pop) {
        e->tp_ns[0] += std::chrono::duration_cast<std::chrono::nanoseconds>(tp1 - tp0).count();
        e->tp_ns[1] += tp_holes;
        e->tp_ns[2] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tp1).count();
        e->tp_calls++;
    }
    e->d.n_live = (int64_t)e->order.size();
    e->order_dirty = true;
    if (!patch) {                                                // the host mirror is authoritative now: close the holes
        compact_host(e);
        set_shard(e);
    }
    return CSF_OK;
}

int csf_set_dest_queue(csf_engine *e, int64_t n, const int32_t *agent, const int64_t *offsets,
                       const double *xyz_stop, int32_t reset) {
    if (!e) return CSF_E_ARG;
    if (n < 0 || (n > 0 && (!agent || !offsets || !xyz_stop))) return fail(e, CSF_E_ARG, "csf_set_dest_queue: bad arguments");
    const int64_t pop = (int64_t)e->order.size();
    // rows the slab must take: every listed queue is written out whole, as it will be after ITS entry of the call (a
    // road user listed twice in an appending call has the rows of its first entry copied again by the second)
    int64_t total = 0;
    std::unordered_map<int32_t, int64_t> grown;                  // appending call: rows of a road user after its last entry
    for (int64_t k = 0; k < n; k++) {
        if (agent[k] < 0 || agent[k] >= pop) return fail(e, CSF_E_ARG, "agent index %d out of range", agent[k]);
        if (offsets[k + 1] < offsets[k]) return fail(e, CSF_E_ARG, "offsets must be non-decreasing");
        if (reset && offsets[k + 1] == offsets[k]) return fail(e, CSF_E_ARG, "reset with an empty queue");
        int64_t rows = offsets[k + 1] - offsets[k];
        if (!reset) {
            auto it = n > 1 ? grown.find(agent[k]) : grown.end();
            rows += it != grown.end() ? it->second : (int64_t)e->h_q[(size_t)e->order[(size_t)agent[k]]].size() / 3;
            if (n > 1) grown[agent[k]] = rows;
        }
        total += rows;
    }
    if (n == 0) return CSF_OK;
    HIPCHK(e, hipSetDevice(e->device));
    // on a current device copy the new queues are appended to the slab and the slots pointed at them
    const bool patch = can_patch_device(e);
    if (patch && e->q_top + (int64_t)(e->pend.rows.size() / 3) + total > e->d.qcap) {
        // (the queues being replaced are still counted among the live ones: room for both)
        int rc = compact_slab(e, total);
        if (rc) return rc;
    }
    if (!patch) {
        int rc = prepare_mutation(e);
        if (rc) return rc;
    }
    for (int64_t k = 0; k < n; k++) {
        const size_t a = (size_t)e->order[(size_t)agent[k]];
        std::vector<double> &qa = e->h_q[a];
        if (reset) {                                             // vehicle.py:642-645
            qa.clear();
            if (reset == 1) e->h_ptr[a] = 0;                     // reset == 2: rows edited in place, pointer kept
        }
        qa.insert(qa.end(), xyz_stop + 3 * offsets[k], xyz_stop + 3 * offsets[k + 1]);  // :646-647
        const int32_t nrows = (int32_t)(qa.size() / 3);
        if (e->h_ptr[a] >= nrows) e->h_ptr[a] = nrows - 1;
        if (!patch) continue;
        const int64_t at_rows = (int64_t)(e->pend.rows.size() / 3);
        e->pend.rows.insert(e->pend.rows.end(), qa.begin(), qa.end());
        if (e->sidx[a].spawn >= 0) {                          // a road user of this batch: its spawn record takes the queue
            SpawnRec &r = e->pend.spawn[(size_t)e->sidx[a].spawn];   // (a new road user's pointer is 0 either way)
            r.qbeg = at_rows;
            r.qlen = nrows;
            continue;
        }
        QueueRec r;
        r.slot = (int32_t)a;
        r.qlen = nrows;
        r.qbeg = at_rows;
        r.mode = reset;
        r.pad = 0;
        if (e->sidx[a].requeue >= 0) {                        // replaced twice in one batch: the last one counts, and a
            QueueRec &old = e->pend.requeue[(size_t)e->sidx[a].requeue];   // rewind requested by either is kept
            if (old.mode == 1) r.mode = 1;
            old = r;
        } else {
            e->sidx[a].requeue = (int32_t)e->pend.requeue.size();
            e->pend.requeue.push_back(r);
        }
    }
    return CSF_OK;
}

int csf_set_road_vertices(csf_engine *e, int32_t n_edges, const int64_t *offsets, const double *xy,
                          const double *F0, const double *sigma) {
    if (!e) return CSF_E_ARG;
    if (n_edges < 0 || (n_edges > 0 && (!offsets || !xy || !F0 || !sigma)))
        return fail(e, CSF_E_ARG, "csf_set_road_vertices: bad arguments");
    HIPCHK(e, hipSetDevice(e->device));
    int rc = prepare_mutation(e);
    if (rc) return rc;
    e->h_road.clear();
    e->road_version++;
    for (int32_t k = 0; k < n_edges; k++)
        for (int64_t v = offsets[k]; v < offsets[k + 1]; v++) {
            e->h_road.push_back(xy[2 * v]);
            e->h_road.push_back(xy[2 * v + 1]);
            e->h_road.push_back(F0[k]);
            e->h_road.push_back(sigma[k]);
        }
    return CSF_OK;
}

int csf_set_incremental(csf_engine *e, int32_t on) {
    if (!e) return CSF_E_ARG;
    e->incremental = on != 0;
    return CSF_OK;
}

int csf_set_params(csf_engine *e, const csf_params *params) {
    if (!e) return CSF_E_ARG;
    int rc = check_params(e, params);
    if (rc) return rc;
    if (params->model != e->d.p.model) return fail(e, CSF_E_ARG, "the model of an engine cannot change");
    if (params->t_s != e->d.p.t_s || params->traj_len != e->d.p.traj_len)
        return fail(e, CSF_E_ARG, "t_s is immutable (parameters.py:516-528)");
    e->d.p = *params;
    derive_consts(e);
    return CSF_OK;
}

int csf_set_param_classes(csf_engine *e, int32_t n_classes, const csf_params *classes) {
    if (!e) return CSF_E_ARG;
    if (n_classes < 1 || n_classes > 256 || !classes) return fail(e, CSF_E_ARG, "csf_set_param_classes: 1 to 256 parameter sets");
    for (int32_t c = 0; c < n_classes; c++) {
        int rc = check_params(e, classes + c);
        if (rc) return rc;
        if (classes[c].t_s != e->d.p.t_s || classes[c].traj_len != e->d.p.traj_len)
            return fail(e, CSF_E_ARG, "parameter set %d: the vehicles of one intersection share t_s (parameters.py:516-528)", c);
    }
    for (int32_t a : e->order)
        if (e->h_cls[(size_t)a] >= n_classes)
            return fail(e, CSF_E_ARG, "a road user still uses parameter set %d", (int)e->h_cls[(size_t)a]);
    if (e->d.hist != nullptr) {                                // the history ring was sized for the present state layout
        int ns = 0;
        for (int32_t c = 0; c < n_classes; c++) ns = std::max(ns, NS_OF[classes[c].model]);
        if (ns != e->d.ns)
            return fail(e, CSF_E_STATE, "parameter sets of a wider vehicle class after csf_enable_history: install the sets first");
    }
    HIPCHK(e, hipSetDevice(e->device));
    int rc = flush_pending(e);                                 // (changes collected for the device assume one parameter set)
    if (rc) return rc;
    const int32_t rule = e->d.p.priority_rule;
    e->classes.assign(classes, classes + n_classes);
    for (csf_params &c : e->classes) c.priority_rule = rule;
    e->d.p = e->classes[0];
    derive_consts(e);
    e->ticks_since_rebin = 1 << 20;                            // binned or not may have changed: decide again (rebin)
    return CSF_OK;
}

int csf_set_agent_class(csf_engine *e, int64_t n, const int32_t *idx, const int32_t *cls) {
    if (!e) return CSF_E_ARG;
    if (n < 0 || (n > 0 && (!idx || !cls))) return fail(e, CSF_E_ARG, "csf_set_agent_class: bad arguments");
    const int64_t pop = (int64_t)e->order.size();
    for (int64_t k = 0; k < n; k++) {
        if (idx[k] < 0 || idx[k] >= pop) return fail(e, CSF_E_ARG, "agent index %d out of range", idx[k]);
        if (cls[k] < 0 || cls[k] >= (int32_t)e->classes.size())
            return fail(e, CSF_E_ARG, "parameter set %d of %d (csf_set_param_classes first)", cls[k], (int)e->classes.size());
    }
    bool device_rows_stale = false;
    for (int64_t k = 0; k < n; k++) {
        const size_t a = (size_t)e->order[(size_t)idx[k]];
        if (e->sidx[a].spawn >= 0) {                          // an arrival still on its way to the device: its spawn record
            e->pend.spawn[(size_t)e->sidx[a].spawn].cls = cls[k];   // carries the set, the patch kernel writes the row and
            e->h_cls[a] = (uint8_t)cls[k];                       // derives the start state with the set's limits
            continue;
        }
        if (e->h_cls[a] != (uint8_t)cls[k]) device_rows_stale = true;
        e->h_cls[a] = (uint8_t)cls[k];
        // A road user that has not moved yet is what its constructor made of it, with the limits of ITS set
        // (vehicle.py:1728-1736).  Only where the host mirror is the state of record: nothing collected for the device
        // (the mirror does not hold what a pending batch will write) and no tick since the last upload.
        if (!e->device_ahead && e->pend.empty() && e->h_ti[a] == 0) {
            side_state(e, a, e->classes[(size_t)cls[k]]);
            e->dirty = true;
        }
    }
    if (device_rows_stale) e->classes_dirty = true;            // (the slots' rows are uploaded with the table)
    return CSF_OK;
}

int csf_set_priority_rule(csf_engine *e, int32_t rule) {
    if (!e) return CSF_E_ARG;
    if (rule < 0 || rule > 1) return fail(e, CSF_E_ARG, "unknown priority rule %d", rule);
    e->d.p.priority_rule = rule;
    derive_consts(e);
    return CSF_OK;
}

int csf_set_v_desired(csf_engine *e, int64_t n, const int32_t *idx, const double *v_desired) {
    if (!e) return CSF_E_ARG;
    if (n < 0 || (n > 0 && (!idx || !v_desired))) return fail(e, CSF_E_ARG, "csf_set_v_desired: bad arguments");
    for (int64_t k = 0; k < n; k++)
        if (idx[k] < 0 || idx[k] >= (int64_t)e->order.size()) return fail(e, CSF_E_ARG, "agent index %d out of range", idx[k]);
    HIPCHK(e, hipSetDevice(e->device));
    for (int64_t k = 0; k < n; k++) {
        const size_t a = (size_t)e->order[(size_t)idx[k]];
        e->h_vdes[a] = v_desired[k];
        // an arrival still on its way to the device: its spawn record carries the desired speed (the patch kernel writes
        // d.vdes[a] from it AFTER the copy below)
        if (e->sidx[a].spawn >= 0) e->pend.spawn[(size_t)e->sidx[a].spawn].vdes = v_desired[k];
    }
    if (!e->dirty) {  // device copy is current: patch it in place
        HIPCHK(e, hipStreamSynchronize(e->main));
        HIPCHK(e, hipMemcpy(e->vdes.p, e->h_vdes.data(), (size_t)e->d.n * sizeof(double), hipMemcpyHostToDevice));
    }
    return CSF_OK;
}

int csf_push_state(csf_engine *e, int64_t n, const int32_t *idx, const double *s) {
    if (!e) return CSF_E_ARG;
    if (n < 0 || (n > 0 && (!idx || !s))) return fail(e, CSF_E_ARG, "csf_push_state: bad arguments");
    for (int64_t k = 0; k < n; k++)
        if (idx[k] < 0 || idx[k] >= (int64_t)e->order.size()) return fail(e, CSF_E_ARG, "agent index %d out of range", idx[k]);
    HIPCHK(e, hipSetDevice(e->device));
    int rc = prepare_mutation(e);
    if (rc) return rc;
    const int ns = e->d.ns;
    const int64_t cap = e->cap;
    for (int64_t k = 0; k < n; k++) {
        const int64_t a = e->order[(size_t)idx[k]];
        for (int c = 0; c < ns; c++) e->h_s[c * cap + a] = s[k * ns + c];
        // keep the model side-state consistent with the pushed vehicle.s
        const int model = e->classes[e->h_cls[(size_t)a]].model;
        if (model == CSF_BALANCINGRIDER) {
            // The integrator restarts from the written vehicle.s (dynamics.py:350-371; roll, steer and yaw keep the winding
            // number they have, the rates are states of vehicle.s); the speed its gains belong to stays (dynamics.py:671-673).
            // (The reference object would ignore the write: BalancingRiderDynamics.step overwrites bicycle.s from its own
            // x and v, dynamics.py:697-704.  Here a written state moves the rider, as for every other class.)
            const double twopi = 6.283185307179586476925286766559;
            auto rewind = [&](double own, double wrapped) { return wrapped + twopi * std::nearbyint((own - wrapped) / twopi); };
            double *x = e->h_lti.data() + a;
            x[0 * cap] = rewind(x[0 * cap], e->h_s[5 * cap + a]);
            x[1 * cap] = rewind(x[1 * cap], -e->h_s[4 * cap + a]);
            x[2 * cap] = e->h_s[7 * cap + a];
            x[3 * cap] = -e->h_s[6 * cap + a];
            x[4 * cap] = rewind(x[4 * cap], -e->h_s[2 * cap + a]);
        } else {
            e->h_ppsi[a] = e->h_s[2 * cap + a];
        }
        if (model == CSF_PLANARBIKE) e->h_lti[a] = e->h_s[4 * cap + a];   // dynamics.x[0] = delta
        const int slot = e->h_ti[a] & (e->d.hist_len - 1);
        e->h_hx[(size_t)slot * cap + a] = e->h_s[a];
        e->h_hy[(size_t)slot * cap + a] = e->h_s[cap + a];
    }
    return CSF_OK;
}

int csf_get_integrator_state(csf_engine *e, double *x, double *psi_unwrapped, uint8_t *zrid) {
    if (!e) return CSF_E_ARG;
    HIPCHK(e, hipSetDevice(e->device));
    int rc = upload_all(e);
    if (rc) return rc;
    rc = csf_sync(e);
    if (rc) return rc;
    const int64_t n = (int64_t)e->order.size();
    if (n == 0) return CSF_OK;
    if (x && (rc = read_rows(e, e->lti.p, 5, x, true))) return rc;
    if (psi_unwrapped && (rc = read_rows(e, e->ppsi.p, 1, psi_unwrapped, false))) return rc;
    if (zrid) {
        std::vector<uint8_t> z((size_t)n);
        if ((rc = read_rows(e, e->zrid.p, 1, z.data(), false))) return rc;
        for (int64_t a = 0; a < n; a++) {
            zrid[2 * a] = z[(size_t)a] != 0;                     // vehicle.py:1949-1950
            zrid[2 * a + 1] = z[(size_t)a] == 0;
        }
    }
    return CSF_OK;
}

int csf_set_integrator_state(csf_engine *e, int64_t n, const int32_t *idx, const double *x, const double *psi_unwrapped,
                             const uint8_t *zrid) {
    if (!e) return CSF_E_ARG;
    if (n < 0 || (n > 0 && !idx)) return fail(e, CSF_E_ARG, "csf_set_integrator_state: bad arguments");
    for (int64_t k = 0; k < n; k++)
        if (idx[k] < 0 || idx[k] >= (int64_t)e->order.size()) return fail(e, CSF_E_ARG, "agent index %d out of range", idx[k]);
    HIPCHK(e, hipSetDevice(e->device));
    int rc = prepare_mutation(e);
    if (rc) return rc;
    const int64_t cap = e->cap;
    for (int64_t k = 0; k < n; k++) {
        const int64_t a = e->order[(size_t)idx[k]];
        if (x)
            for (int c = 0; c < 5; c++) e->h_lti[c * cap + a] = x[k * 5 + c];
        if (psi_unwrapped) e->h_ppsi[a] = psi_unwrapped[k];
        if (zrid) e->h_zrid[a] = zrid[2 * k] ? 1 : 0;
    }
    return CSF_OK;
}

// One tick, all on the main stream (dependent launches in one stream cost ~2 us; a cross-stream event wait was
// measured at ~11 us here, more than running the destination-force phase beside the pair kernel saves):
//   main:  bounds - pair - road - agent(DEST|COMBINE|INTEGRATE) - [all-gather(records)]       (world > 1: RCCL)
// CSF_COMM_STREAM=second moves the collective to a second stream and issues the destination-force phase of the
// next tick (it needs only the agent's own state) before the wait on it:
// The pair term of a tick: one launch, or - class-segmented order, several parameter sets - one launch of the culling kernel
// per set's run of places, with that set's constants, field (vehicle class) and far-field radius; `base` carries what the
// caller wants of every launch (counters, next tick's circles or not).  The optional time stamps bracket the first launch.
static void launch_pair_all(csf_engine *e, const Dev &base, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr) {
    if (e->segs.empty()) {
        launch_pair(base, e->main, t0, t1);
        return;
    }
    const bool one_grid = e->seg_total_by > 0 && base.pair_count == nullptr && base.dyn_recv && e->knobs.seg_grid != 0;
    if (one_grid) {      // the runs of all TwoD-field sets in one grid
        Dev dd = base;
        dd.n_classes = 1;
        dd.recv_binned = 0;
        dd.segtab = e->segtab.p;
        dd.n_seg = (int32_t)e->h_segtab.size();
        launch_pair_segments(dd, e->seg_total_by, e->main, t0, t1);
        t0 = t1 = nullptr;
    }
    for (const csf_engine::Segment &sg : e->segs) {
        if (one_grid && e->classes[(size_t)sg.cls].model != CSF_BICYCLE) continue;   // (done above; the Bicycle field has a kernel of its own)
        Dev dd = base;
        dd.p = e->classes[(size_t)sg.cls];
        dd.p.priority_rule = base.p.priority_rule;
        dd.pc = sg.pc;
        dd.n_classes = 1;
        dd.src_beg = sg.beg;
        dd.n_src = sg.end;
        dd.chunk_units = sg.chunk_units;
        dd.n_split = sg.n_split;
        dd.part_base = sg.part_base;
        dd.recv_binned = 0;
        launch_pair(dd, e->main, t0, t1);
        t0 = t1 = nullptr;
    }
}

// Which order for the collective?  Timed on the communicator as it is - its ranks, its links - with launches that change
// nothing: the pair kernel on the records as they are (it writes partial sums and nothing else here) followed by the
// all-gather of the records (in place: what arrives is what was there), 8 + 32 times in stream order and 8 + 32 times with
// the collective on the second stream behind an event and a small kernel of the next tick in front of the wait.  The two
// times are max-reduced over the ranks, so that every rank keeps the same order.
static int calibrate_comm_stream_body(csf_engine *e);

// (the flag is set only when the measurement went through: after an error the order is the plain one - collective in stream
// order - and the next csf_step tries again)
int calibrate_comm_stream(csf_engine *e) {
    const int rc = calibrate_comm_stream_body(e);
    if (rc) e->comm_second = false;
    else e->comm_calibrated = true;
    return rc;
}

static int calibrate_comm_stream_body(csf_engine *e) {
    if (!e->nccl || e->loopback) {
        e->comm_second = e->knobs.comm_second > 0;
        return CSF_OK;
    }
    {   // CSF_COMM_STREAM decides which collectives a rank issues below: every rank must have read the same value, or the
        // ranks that measure wait for the ones that do not.  Agree on it before branching (max and min over the ranks).
        float v[2] = {(float)e->knobs.comm_second, -(float)e->knobs.comm_second};
        DevBuf<float> t;
        HIPCHK(e, t.alloc(2));
        HIPCHK(e, hipMemcpy(t.p, v, sizeof v, hipMemcpyHostToDevice));
        NCCLCHK(e, g_rccl.AllReduce(t.p, t.p, 2, ncclFloat32, ncclMax, e->nccl, e->main));
        HIPCHK(e, hipStreamSynchronize(e->main));
        HIPCHK(e, hipMemcpy(v, t.p, sizeof v, hipMemcpyDeviceToHost));
        t.release();
        if (v[0] != -v[1]) return fail(e, CSF_E_STATE, "CSF_COMM_STREAM differs between the ranks of this communicator: set it on all of them or on none");
    }
    if (e->knobs.comm_second >= 0) {
        e->comm_second = e->knobs.comm_second > 0;
        return CSF_OK;
    }
    {
        int rc = set_fov_band(e);
        if (rc) return rc;
    }
    Dev dd = e->d;
    dd.edge = nullptr;          // (no per-agent launch follows that would take undecided pairs over)
    dd.bnd_next = nullptr;      // (the circles stay as they are)
    struct Events {   // (destroyed on every way out)
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() {
            if (a) (void)hipEventDestroy(a);
            if (b) (void)hipEventDestroy(b);
        }
    } evs;
    HIPCHK(e, hipEventCreate(&evs.a));
    HIPCHK(e, hipEventCreate(&evs.b));
    const hipEvent_t a = evs.a, b = evs.b;
    float us[2] = {0.f, 0.f};
    for (int mode = 0; mode < 2; mode++) {
        e->comm_second = mode == 1;
        for (int it = 0; it < 40; it++) {
            if (it == 8) HIPCHK(e, hipEventRecord(a, e->main));
            if (mode == 1) {
                if (dd.recs_valid) launch_sorted_copy(dd, e->main);   // (stands for the destination-force phase of the next tick)
                int rc = wait_gather(e);
                if (rc) return rc;
            }
            if (dd.n_live > 1 && dd.hi > dd.lo) launch_pair_all(e, dd, nullptr, nullptr);
            if (mode == 1) HIPCHK(e, hipEventRecord(e->ev_integ, e->main));
            int rc = all_gather_records(e);
            if (rc) return rc;
        }
        int rc = wait_gather(e);
        if (rc) return rc;
        HIPCHK(e, hipEventRecord(b, e->main));
        HIPCHK(e, hipStreamSynchronize(e->main));
        if (e->comm) HIPCHK(e, hipStreamSynchronize(e->comm));
        float ms = 0.f;
        HIPCHK(e, hipEventElapsedTime(&ms, a, b));
        us[mode] = ms * 1e3f / 32.f;
    }
    {   // the slowest rank's figures, on every rank
        DevBuf<float> t;
        HIPCHK(e, t.alloc(2));
        HIPCHK(e, hipMemcpy(t.p, us, sizeof us, hipMemcpyHostToDevice));
        NCCLCHK(e, g_rccl.AllReduce(t.p, t.p, 2, ncclFloat32, ncclMax, e->nccl, e->main));
        HIPCHK(e, hipStreamSynchronize(e->main));
        HIPCHK(e, hipMemcpy(us, t.p, sizeof us, hipMemcpyDeviceToHost));
        t.release();
    }
    e->comm_cal_us[0] = us[0];
    e->comm_cal_us[1] = us[1];
    e->comm_second = us[1] < us[0];
    return CSF_OK;
}

// ---- mid-size populations: the whole tick in one launch (csf_mid.hip) --------------------------------------------------
// One device, one parameter set, the plain all-pairs kernel's range (below PLAIN_BELOW road users; up to SMALL_MAX the one-wave
// kernel is faster still), nothing that wants the two kernels apart (time stamps per kernel, wave traces, pair counters).
// Where two launches take over again (tools/mid_rate.py, profiles/r5_v2_models_mid_rate.jsonl; microseconds per tick, one launch /
// two): the classes with twelve waves per workgroup and the cull-first pair sums win up to ~2 100 road users (2 048: TwoD 20.9 /
// 22.0, PlanarPoint 20.6 / 21.7), the InvPendulum (eight waves) and the Bicycle field (four receivers per item, no cull) only
// to ~1 300 (1 024: 18.2 / 19.2 and 13.7 / 14.0; 2 048: 29.2 / 26.0 and 20.0 / 17.5).
static int64_t mid_below_for(const csf_engine *e) {
    if (e->knobs.mid_below > 0) return e->knobs.mid_below;
    return e->d.p.model == CSF_INVPEND || e->d.p.model == CSF_BICYCLE ? 1280 : 2176;
}

static bool mid_fused_ok(const csf_engine *e) {
    const Dev &d = e->d;
    return e->knobs.fused_mid != 0 && e->knobs.pair_variant < 0 && d.pair_variant == 1 && d.n_live > 1 && d.n_live < mid_below_for(e) && e->classes.size() == 1 &&
           d.p.model != CSF_UNCONTROLLED && d.p.model != CSF_BALANCINGRIDER && e->world == 1 && !e->nccl && !e->loopback && e->knobs.fake_world <= 1 && e->profile <= 0 &&
           d.trace == nullptr && d.pair_count == nullptr && d.lo == 0 && d.hi == d.n && e->segs.empty() && e->state_all_current &&
           d.src_beg == 0 && d.n_src / 64 * 8 <= 384;                  // (csf_mid.hip: MID_ITEMS_MAX items of the largest group)
}

// the other halves of the double buffers <- this tick's records (sentinels of free and padding slots included) and state
// the second halves of the double buffers (allocations synchronise: never inside something that is being timed)
static int alt_alloc(csf_engine *e) {
    const size_t nrec = e->rec.n;
    if (e->rec_alt.n >= nrec) return CSF_OK;
    HIPCHK(e, e->rec_alt.alloc(nrec));
    HIPCHK(e, e->recg_alt.alloc(e->recg.n));
    HIPCHK(e, e->rec2_alt.alloc(e->rec2.n));
    HIPCHK(e, e->src64_a.alloc(3 * (size_t)e->cap));
    HIPCHK(e, e->src64_b.alloc(3 * (size_t)e->cap));
    HIPCHK(e, e->recs_alt.alloc(e->recs.n));
    // the permanent sentinel (alloc_all: the last record, no slot's; the padding of the class-segmented order points at it)
    // exists in BOTH halves: after an odd number of one-launch ticks d.rec is the other half, and a population that then
    // outgrows this path would read (0, 0, 0, 0) - a road user at the origin - where its order is padded
    HIPCHK(e, hipMemcpyAsync(e->rec_alt.p + e->sent_slot, e->rec.p + e->sent_slot, sizeof(float4), hipMemcpyDeviceToDevice, e->main));
    HIPCHK(e, hipMemcpyAsync(e->rec2_alt.p + e->sent_slot, e->rec2.p + e->sent_slot, sizeof(float2), hipMemcpyDeviceToDevice, e->main));
    if ((size_t)e->sent_slot < e->recg.n)
        HIPCHK(e, hipMemcpyAsync(e->recg_alt.p + e->sent_slot, e->recg.p + e->sent_slot, sizeof(float4), hipMemcpyDeviceToDevice, e->main));
    return CSF_OK;
}

static int mid_sync(csf_engine *e, unsigned *cnt = nullptr, int64_t ncnt = 0, unsigned *through = nullptr) {
    Dev &d = e->d;
    {
        int rca = alt_alloc(e);
        if (rca) return rca;
    }
    float4 *rec_o = d.rec == e->rec.p ? e->rec_alt.p : e->rec.p, *recg_o = d.recg == e->recg.p ? e->recg_alt.p : e->recg.p;
    float4 *recs_o = d.recs == e->recs.p ? e->recs_alt.p : e->recs.p;
    float2 *rec2_o = d.rec2 == e->rec2.p ? e->rec2_alt.p : e->rec2.p;
    double *cur = e->mid_cur_is_a ? e->src64_a.p : e->src64_b.p;
    // (one launch: six copy / fill calls cost a launch and its gap each, once per re-binning)
    launch_chase_sync(d, rec_o, recg_o, recs_o, rec2_o, cur, (int64_t)std::min((size_t)d.n_pad, e->recg.n), cnt, ncnt, through, e->main);
    HIPCHK(e, hipGetLastError());
    if (cnt == nullptr) e->chase_resume = false;                  // (the one-launch tick's call: the arrival counters are as they were)
    e->mid_synced = true;
    return CSF_OK;
}

// (behind bounds_before_pair - the re-binning: new origins of the precise records, and a new d.n_src, which the gate reads)
static int enqueue_mid_tick(csf_engine *e) {
    Dev &d = e->d;
    int rc;
    if ((rc = set_fov_band(e))) return rc;
    if (!e->mid_synced && (rc = mid_sync(e))) return rc;
    if (d.nv > 0) launch_road(d, e->main);
    float4 *rec_o = d.rec == e->rec.p ? e->rec_alt.p : e->rec.p, *recg_o = d.recg == e->recg.p ? e->recg_alt.p : e->recg.p;
    float2 *rec2_o = d.rec2 == e->rec2.p ? e->rec2_alt.p : e->rec2.p;
    double *cur = e->mid_cur_is_a ? e->src64_a.p : e->src64_b.p, *nxt = e->mid_cur_is_a ? e->src64_b.p : e->src64_a.p;
    Dev dd = d;
    dd.rec_w = rec_o;
    dd.recg_w = recg_o;
    dd.rec2_w = rec2_o;
    dd.src64 = cur;
    dd.src64_w = nxt;
    // road users per workgroup: about one workgroup (of eight waves, at the per-agent code's 255 registers a CU holds one) per CU
    int G = e->knobs.mid_group;
    if (G != 4 && G != 8 && G != 16 && G != 32) {
        G = 4;
        while (G < 32 && (d.hi - d.lo + G - 1) / G > 256) G *= 2;
    }
    dd.mid_group = G;
    // (nothing has traded places yet: a refused launch leaves the engine where it was)
    if (!launch_mid_tick(dd, e->main)) return fail(e, CSF_E_STATE, "the one-launch tick does not take this population (%lld sources, groups of %d)", (long long)d.n_src, G);
    HIPCHK(e, hipGetLastError());
    // the halves trade places: what the launch wrote is what every later launch reads
    d.rec = d.rec_w = rec_o;
    d.recg = d.recg_w = recg_o;
    d.rec2 = d.rec2_w = rec2_o;
    e->mid_cur_is_a = !e->mid_cur_is_a;
    bounds_after_pair(e, true);
    e->moves++;
    e->mid_ticks++;
    d.tick++;
    return CSF_OK;
}

// ---- the per-agent launch inside the pair launch's drain (csf_dev.h: chase_cnt) ---------------------------------------------------
// At the headline size the pair launch idles most CUs for its last ~24 us while the per-agent launch (8 us + two launch gaps) waits
// behind it.  Here the two run side by side on the engine's two streams, which trade places every tick:
//   tick t:      P: pair(t)                      Q: gate(t) - agent(t)     [agent(t) waits, per 64 slots, for pair(t)'s arrivals]
//   tick t + 1:  Q: pair(t + 1)  (behind agent(t): every record is in place)      P: gate(t + 1) - agent(t + 1)  (behind pair(t))
// No event between the streams in steady state: a kernel only ever waits (on the device) for one that was enqueued BEFORE it, on
// a stream whose earlier work does not wait for it - so whatever the runtime maps the streams to, the grid drains (mapped to one
// hardware queue the launches simply serialise).  Next tick's records go to the other halves of the double buffers, hand-overs
// read the fp64 snapshot of the tick's start (as in csf_mid.hip).  The streams meet (two events) when the path is left.
static bool rebin_due(const csf_engine *e) {   // (bounds_before_pair's condition: this tick renews the binned order, on the main stream)
    return e->ticks_since_rebin + e->moved_unbinned >= e->knobs.rebin_ticks || e->ticks_since_rebin * e->churn >= e->knobs.rebin_churn;
}

// could ticks of this engine take the path at all (what does not change from tick to tick)?
static bool chase_shape(const csf_engine *e) {
    const Dev &d = e->d;
    const int m = d.p.model;
    return e->knobs.chase != 0 && (m == CSF_TWOD || m == CSF_INVPEND || m == CSF_PLANARPOINT) && e->classes.size() == 1 && d.pair_variant == 0 && d.classify &&
           !d.recv_binned && e->world == 1 && !e->nccl && !e->loopback && d.nv == 0 && d.hist == nullptr && e->comm != nullptr;   // (CSF_FAKE_SHARD: a rank's block of receivers, for tools/fake_shard.py)
}

// everything but "no re-binning this tick"
static bool chase_eligible(const csf_engine *e, int64_t ticks_left) {
    const Dev &d = e->d;
    if (e->knobs.chase == 0) return false;
    const int m = d.p.model;
    if (m != CSF_TWOD && m != CSF_INVPEND && m != CSF_PLANARPOINT) return false;
    const bool rebin_now = false;
    return e->classes.size() == 1 && d.n_classes == 1 && d.pair_variant == 0 && d.classify && d.recs_valid && !d.recv_binned && d.dyn_recv &&
           (d.rpb == 32 || d.rpb == 16 || d.rpb == 8) && d.n_split <= 16 && e->world == 1 && !e->nccl && !e->loopback &&
           d.nv == 0 && d.hist == nullptr && d.pair_count == nullptr && e->segs.empty() && e->state_all_current &&   // (wave traces allowed: tools/chase_timeline.py)
           e->pend.empty() && !rebin_now && !e->bound_stale && e->bounds_fresh && d.n_live > 1 && d.hi > d.lo && (d.lo & 63) == 0 && d.replay_len == nullptr &&
           ((d.lo == 0 && d.hi == d.n) || e->knobs.fake_world > 1) &&
           e->comm != nullptr && (e->chase_prev || ticks_left >= 4);
}

static bool chase_ok(const csf_engine *e, int64_t ticks_left) { return !rebin_due(e) && chase_eligible(e, ticks_left); }

static int chase_join(csf_engine *e);
static int alt_alloc(csf_engine *e);

// what the side-by-side tick needs beside the engine's arrays (allocations synchronise: before a measurement, not inside it)
static int chase_alloc(csf_engine *e) {
    const size_t nw = (size_t)((e->d.hi - e->d.lo + 63) / 64);
    HIPCHK(e, e->chase_cnt.reserve(nw + 64));
    if (e->chase_misc.n == 0) {
        HIPCHK(e, e->chase_misc.reserve(64));
        // once per engine: the runtime makes a stream's hardware queue when the stream is first used, and loads a kernel's code when it
        // is first asked for - a hundred microseconds and more each, which would land in the first side-by-side tick (and, for a
        // caller who times twenty ticks, in the figure)
        HIPCHK(e, hipMemsetAsync(e->chase_misc.p, 0, 64 * sizeof(unsigned), e->comm));
        preload_chase_kernels();
        // ... and a queue gets its scratch memory when a kernel that spills is first dispatched on it: one pair launch that hands
        // nothing over (as csf_count_pairs's: partial sums and next tick's circles are written again by the tick's own launch)
        // and a kernel that asks for the per-agent kernels' bytes per lane, on the second stream
        if (e->d.n_live > 1 && e->d.hi > e->d.lo && !e->dirty_layout_for_warm()) {
            Dev dw = e->d;
            dw.edge = nullptr;
            dw.pair_count = nullptr;
            dw.trace = nullptr;
            launch_pair(dw, e->comm);
        }
        launch_chase_scratch_warm(e->chase_misc.p, e->comm);
        HIPCHK(e, hipGetLastError());
        HIPCHK(e, hipStreamSynchronize(e->comm));
    }
    return alt_alloc(e);
}

// What an engine measured holds for the next engine of the same kind in this process (same device, same rider class, same size to a
// factor of two): bench.py's timed engine takes over what its scratch engine found, and a caller that steps a few ticks per call
// - too few for a measurement of its own - is served by an earlier one.  0: nothing measured yet.
static std::atomic<int> g_chase_found[8][8][32];
static std::atomic<int> *chase_found_slot(const csf_engine *e) {
    int b = 0;
    for (int64_t n = std::max<int64_t>(e->d.n, 1); n > 1 && b < 31; n >>= 1) b++;
    return &g_chase_found[e->device & 7][e->d.p.model & 7][b];
}

// the measurement's last event has been reached: decide
static void chase_cal_resolve(csf_engine *e, bool wait) {
    if (e->cal_phase != 4) return;
    if (wait ? hipEventSynchronize(e->cal_ev[3]) != hipSuccess : hipEventQuery(e->cal_ev[3]) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    float t[3] = {0, 0, 0};
    bool ok = true;
    for (int k = 0; k < 3; k++) ok = ok && hipEventElapsedTime(&t[k], e->cal_ev[k], e->cal_ev[k + 1]) == hipSuccess && e->cal_ticks[k] > 0;
    if (ok) {
        for (int k = 0; k < 3; k++) t[k] = t[k] * 1e3f / (float)e->cal_ticks[k];        // microseconds per tick of every period
        e->chase_cal_us[0] = 0.5 * (double)(t[0] + t[2]);
        e->chase_cal_us[1] = (double)t[1];
        // (the side-by-side period against the MEAN of the in-turn periods on either side of it: a clock that is still ramping up
        // cancels; by a per cent, so that noise does not decide)
        e->chase_state = t[1] < 0.99f * 0.5f * (t[0] + t[2]) ? 1 : -1;
        chase_found_slot(e)->store(e->chase_state);
    }
    (void)hipGetLastError();
    e->cal_phase = 0;
}

// This tick beside the pair launch?  CSF_CHASE=1 drives the one-off measurement: three whole periods between re-binnings - in turn, side
// by side, in turn -, each from its re-binning tick to the next, so that what entering and leaving the side-by-side path costs once
// per period (the halves made equal, the streams meeting) is part of what is compared.
static bool chase_take(csf_engine *e, int64_t ticks_left) {
    const bool ok = chase_ok(e, ticks_left);
    if (e->knobs.chase != 1 || e->chase_state != 0) return ok && (e->knobs.chase >= 2 || e->chase_state == 1);
    if (e->cal_phase == 0) {
        const int found = chase_found_slot(e)->load();                     // an earlier engine of this kind has measured
        if (found != 0) {
            e->chase_state = found;
            return ok && found == 1;
        }
    }
    if (e->cal_phase == 4) {
        chase_cal_resolve(e, false);
        return ok && e->chase_state == 1;
    }
    if (rebin_due(e) && chase_eligible(e, ticks_left)) {                   // a period ends, the next begins (this tick: the launches in turn)
        if (e->cal_phase == 0) {
            // three periods inside this call, on clocks that have had time to come up (a fresh process finds the device in a low
            // power state: its first few hundred ticks run up to 1.7 x slower and speed up as they go), nothing else sampled
            if (ticks_left < 3 * e->knobs.rebin_ticks + 2 || e->d.tick < 256 || e->profile > 0) return false;
            for (hipEvent_t &ev : e->cal_ev)
                if (!ev && hipEventCreate(&ev) != hipSuccess) return false;
            if (chase_alloc(e) != CSF_OK) return false;                     // (now, not inside the side-by-side period)
            if (hipEventRecord(e->cal_ev[0], e->main) != hipSuccess) return false;
            e->cal_phase = 1;
            for (int64_t &c : e->cal_ticks) c = 0;
        } else {
            if ((e->cal_phase == 2 && chase_join(e) != CSF_OK) || hipEventRecord(e->cal_ev[e->cal_phase], e->main) != hipSuccess) {
                e->cal_phase = 0;
                return false;
            }
            e->cal_phase++;
        }
        if (e->cal_phase <= 3) e->cal_ticks[e->cal_phase - 1]++;
        return false;
    }
    if (e->cal_phase >= 1 && e->cal_phase <= 3) {
        if (!ok) {                                                          // (something else got in the way: start over later)
            e->cal_phase = 0;
            return false;
        }
        e->cal_ticks[e->cal_phase - 1]++;
        return e->cal_phase == 2;
    }
    return false;
}

// the two streams meet: whatever follows runs on the main stream alone
static int chase_join(csf_engine *e) {
    if (!e->chase_prev) return CSF_OK;
    // (one event: the main stream behind the second; the second stream is put behind the main one when the path is entered again)
    HIPCHK(e, hipEventRecord(e->ev_gather, e->comm));
    HIPCHK(e, hipStreamWaitEvent(e->main, e->ev_gather, 0));
    e->chase_prev = false;
    return CSF_OK;
}

static int enqueue_chase_tick(csf_engine *e, csf_engine::ProfSlot *ps, csf_engine::ProfSlot *po) {
    Dev &d = e->d;
    int rc;
    if ((rc = bounds_before_pair(e))) return rc;                 // (chase_ok: no re-binning, circles in place - nothing is launched)
    if ((rc = set_fov_band(e))) return rc;
    if (!e->chase_prev) {                                         // entering: the second stream behind everything so far
        if ((rc = chase_alloc(e))) return rc;
        // Nothing but side-by-side ticks since the halves were made equal (the last call ended with such ticks and nothing ran in
        // between): counters, round and stream parity carry on, and entering costs one event.  Else one launch makes the halves
        // equal and clears the arrival counters and the gate's counter (the error word stays).
        if (!e->mid_synced || !e->chase_resume) {
            if ((rc = mid_sync(e, e->chase_cnt.p, (int64_t)e->chase_cnt.n, e->chase_misc.p))) return rc;
            e->chase_round = 0;
            e->chase_parity = 0;
        }
        e->chase_resume = true;
        HIPCHK(e, hipEventRecord(e->ev_integ, e->main));
        HIPCHK(e, hipStreamWaitEvent(e->comm, e->ev_integ, 0));
    }
    hipStream_t P = e->chase_parity ? e->comm : e->main, Q = e->chase_parity ? e->main : e->comm;
    float4 *rec_o = d.rec == e->rec.p ? e->rec_alt.p : e->rec.p, *recg_o = d.recg == e->recg.p ? e->recg_alt.p : e->recg.p;
    float4 *recs_o = d.recs == e->recs.p ? e->recs_alt.p : e->recs.p;
    float2 *rec2_o = d.rec2 == e->rec2.p ? e->rec2_alt.p : e->rec2.p;
    double *cur = e->mid_cur_is_a ? e->src64_a.p : e->src64_b.p, *nxt = e->mid_cur_is_a ? e->src64_b.p : e->src64_a.p;
    const int64_t groups = (d.hi - d.lo + d.rpb - 1) / d.rpb, wgs = groups * d.n_split;
    e->chase_round++;
    Dev dd = d;
    dd.rec_w = rec_o;
    dd.recg_w = recg_o;
    dd.rec2_w = rec2_o;
    dd.recs_w = recs_o;
    dd.src64 = cur;
    dd.src64_w = nxt;
    dd.chase_cnt = e->chase_cnt.p;
    dd.chase_misc = e->chase_misc.p;
    dd.chase_round = e->chase_round;
    dd.chase_clock = nullptr;
    if (!e->knobs.chase_clock.empty()) {   // (a ring of 128 ticks, initialised once: the tool reads fewer than that)
        if (e->chase_clock.n == 0) {
            HIPCHK(e, e->chase_clock.alloc(128 * 8));
            std::vector<unsigned long long> init(128 * 8, 0ull);
            for (int r = 0; r < 128; r++) init[8 * r + 0] = init[8 * r + 4] = ~0ull;       // (minima start high)
            HIPCHK(e, hipMemcpy(e->chase_clock.p, init.data(), init.size() * sizeof(init[0]), hipMemcpyHostToDevice));
        }
        dd.chase_clock = e->chase_clock.p;
        dd.chase_slot = (uint32_t)(e->chase_ticks & 127);
    }
    // (cumulative: the counter is cleared when the path is entered)
    dd.chase_gate = (uint32_t)((int64_t)(e->chase_round - 1) * wgs + wgs * e->knobs.chase_gate_pct / 100);
    launch_pair(dd, P, ps ? ps->ev[0] : nullptr, ps ? ps->ev[1] : nullptr);
    if (ps) ps->pair = true;
    bounds_after_pair(e, true);
    if (!launch_agent_chase(dd, Q, po ? po->ev[4] : nullptr, po ? po->ev[5] : nullptr))
        return fail(e, CSF_E_STATE, "the per-agent kernel beside the pair launch is not built for vehicle class %d", (int)d.p.model);
    if (po) po->agent = true;
    HIPCHK(e, hipGetLastError());
    // the halves trade places: what the launch wrote is what every later launch reads
    d.rec = d.rec_w = rec_o;
    d.recg = d.recg_w = recg_o;
    d.rec2 = d.rec2_w = rec2_o;
    d.recs = d.recs_w = recs_o;
    e->mid_cur_is_a = !e->mid_cur_is_a;
    e->chase_parity ^= 1;
    e->chase_prev = true;
    e->chase_ticks++;
    e->moves++;
    d.tick++;
    return CSF_OK;
}

//   main:  agent(DEST) - wait(ev_gather) - bounds - pair - road - agent(COMBINE|INTEGRATE) - record(ev_integ)
//   comm:  wait(ev_integ) - all-gather(records) - record(ev_gather)
static int enqueue_tick(csf_engine *e, int64_t ticks_left = 1) {
    Dev &d = e->d;
    if (chase_take(e, ticks_left)) {                              // the per-agent launch beside the pair launch (large populations)
        int rcp = CSF_OK;
        csf_engine::ProfSlot *ps = prof_slot(e, &rcp);
        if (rcp) return rcp;
        return enqueue_chase_tick(e, ps, (ps && ((e->prof_ticks++ % 8 == 0) || !e->knobs.chase_clock.empty())) ? ps : nullptr);
    }
    {
        int rcj = chase_join(e);
        if (rcj) return rcj;
    }
    bool bounds_done = false;
    if (mid_fused_ok(e)) {
        // the re-binning inside may move d.n_src past what the one-launch tick takes: ask again behind it, and carry on with
        // two launches (the bounds are in place) rather than lose the tick
        int rcb = bounds_before_pair(e);
        if (rcb) return rcb;
        bounds_done = true;
        if (mid_fused_ok(e)) return enqueue_mid_tick(e);
    }
    e->mid_synced = false;
    const bool sharded = e->world > 1 || e->nccl != nullptr || e->loopback;  // a 1-rank communicator rehearses the sharded path
    int rc = CSF_OK;
    csf_engine::ProfSlot *ps = prof_slot(e, &rc);
    if (rc) return rc;
    // time stamps cost a few microseconds of launch gap per kernel: the pair kernel (what the roofline is computed from)
    // takes them on every sampled tick, the other kernels on every 8th of those
    csf_engine::ProfSlot *po = (ps && (e->prof_ticks++ % 8 == 0)) ? ps : nullptr;
    const bool overlap = sharded && !e->loopback && comm_second_stream(e);
    if (overlap) {
        launch_agent(d, PH_DEST, e->main);
        if ((rc = wait_gather(e))) return rc;
    }
    if (!bounds_done && (rc = bounds_before_pair(e))) return rc;
    // sharded: the other ranks' records arrived in index order; a coalesced tile fill from the binned copy saves the
    // pair kernel 5 - 7 us at every shard size, the copy costs ~3 us
    if (sharded && d.recs_valid && e->ticks_since_rebin > 1) {
        launch_sorted_copy(d, e->main, e->xbuf_fresh);
        e->xbuf_fresh = false;
    }
    if ((rc = set_fov_band(e))) return rc;
    if (d.n_live > 1 && d.hi > d.lo) {
        launch_pair_all(e, d, ps ? ps->ev[0] : nullptr, ps ? ps->ev[1] : nullptr);
        if (ps) ps->pair = true;
    }
    bounds_after_pair(e, true);
    if (d.nv > 0 && d.hi > d.lo) {
        launch_road(d, e->main, po ? po->ev[2] : nullptr, po ? po->ev[3] : nullptr);
        if (po) po->road = true;
    }
    launch_agent(d, overlap ? (PH_COMBINE | PH_INTEGRATE) : (PH_DEST | PH_COMBINE | PH_INTEGRATE), e->main,
                 po ? po->ev[4] : nullptr, po ? po->ev[5] : nullptr);
    // a rank integrates its own block only: the others' fp64 state is stale from now on (a 1-rank communicator owns every block)
    if (e->world > 1 || e->loopback) e->state_all_current = false;
    e->moves++;
    if (po) po->agent = true;
    HIPCHK(e, hipGetLastError());
    d.tick++;
    if (sharded && !e->loopback) {
        if (overlap) HIPCHK(e, hipEventRecord(e->ev_integ, e->main));
        hipStream_t cs = overlap ? e->comm : e->main;
        if (po) HIPCHK(e, hipEventRecord(po->ev[6], cs));
        if ((rc = all_gather_records(e))) return rc;
        if (po) {
            HIPCHK(e, hipEventRecord(po->ev[7], cs));
            po->gather = true;
        }
    }
    return CSF_OK;
}

// the loopback "all-gather": every member's own record block is copied into the record arrays of the others
static int loopback_exchange(csf_engine *const *g, int world) {
    for (int r = 0; r < world; r++) {
        csf_engine *src = g[r];
        const size_t shard = (size_t)(src->d.n_pad / world);
        for (int p = 0; p < world; p++) {
            if (p == r) continue;
            // (one block of exchange records per peer, where a real run has one collective: csf_dev.h xbuf)
            if (!g[p]->xbuf.p || !src->xbuf.p) return fail(g[p], CSF_E_DEVICE, "no exchange buffer (allocation failed when the shard layout was set)");
            HIPCHK(g[p], hipMemcpyAsync(g[p]->xbuf.p + 2 * r * shard, src->xbuf.p + 2 * r * shard, 2 * shard * sizeof(float4),
                                        hipMemcpyDeviceToDevice, src->main));
            g[p]->xbuf_fresh = true;
        }
    }
    return CSF_OK;
}

// A handful of road users of one class (not the UncontrolledVehicle's) on one device, nothing sampled or recorded per tick: the whole tick in one
// wave, all ticks of the call in one launch (csf_agent.hip: small_tick_kernel).  A pinned pair-kernel variant (the test suite's
// CSF_PAIR_VARIANT) keeps the general path.
static bool small_road_ok(const Dev &d) {   // no road, or a small one: staged in LDS, at most 256 vertices per lane and tick
    if (d.nv == 0) return true;
    int64_t P = 1;
    while (P < d.n) P <<= 1;
    return d.rg_nx == 0 && d.nv_pad <= SMALL_ROAD_MAX && d.nv_pad * P <= 256 * WAVE;
}

static bool small_fused_ok(const csf_engine *e) {
    const Dev &d = e->d;
    const int m = d.p.model;
    return e->knobs.fused_small != 0 && e->knobs.pair_variant < 0 && d.n >= 1 && d.n <= SMALL_MAX && d.n_live == d.n &&
           e->classes.size() == 1 && m != CSF_UNCONTROLLED && small_road_ok(d) &&
           e->world == 1 && !e->nccl && !e->loopback && e->knobs.fake_world <= 1 && d.hist == nullptr && e->profile <= 0 &&
           d.atrace == nullptr && d.lo == 0 && d.hi == d.n && e->pend.empty() && !e->dirty;
}

// the mapped host buffer of the packed read-back (csf_get_tick, csf_step_get_tick), large enough for the population
static int snap_reserve(csf_engine *e, size_t need) {
    if (need <= e->snap_bytes) return CSF_OK;
    if (e->snap_host) {
        HIPCHK(e, hipStreamSynchronize(e->main));
        HIPCHK(e, hipHostFree(e->snap_host));
        e->snap_host = nullptr;
        e->snap_bytes = 0;
    }
    const size_t want = std::max<size_t>(need * 2, 4096);
    HIPCHK(e, hipHostMalloc(&e->snap_host, want, hipHostMallocMapped));
    HIPCHK(e, hipHostGetDevicePointer((void **)&e->snap_dev, e->snap_host, 0));
    e->snap_bytes = want;
    return CSF_OK;
}

static int step_impl(csf_engine *e, int64_t n_ticks, bool want_snap, bool *snapped) {
    if (snapped) *snapped = false;
    if (!e) return CSF_E_ARG;
    if (n_ticks < 0) return fail(e, CSF_E_ARG, "n_ticks must be >= 0");
    HIPCHK(e, hipSetDevice(e->device));
    if (e->loopback) return fail(e, CSF_E_STATE, "members of a loopback group are stepped with csf_step_group");
    if (e->world > 1 && !e->nccl) return fail(e, CSF_E_STATE, "csf_comm_init must run before csf_step when world > 1");
    int rc = upload_all(e);
    if (rc) return rc;
    if (e->order.empty()) {  // intersection.py:888: nothing to do, time still advances
        e->d.tick += n_ticks;
        return CSF_OK;
    }
    if (n_ticks > 0 && !e->comm_calibrated && (rc = calibrate_comm_stream(e))) return rc;
    if (n_ticks > 0 && small_fused_ok(e)) {
        if ((rc = set_fov_band(e))) return rc;                 // (the bands of the fp32 decisions: tracked_precise, side_undecided)
        // csf_step_get_tick: the kernel packs the read-back itself behind its last tick, when slots are the population order
        bool snap = false;
        if (want_snap) {
            if ((rc = sync_order(e))) return rc;
            const size_t n = e->order.size();
            snap = e->d.order == nullptr && snap_reserve(e, n * ((size_t)(e->d.ns + 2) * sizeof(double) + sizeof(int32_t) + 3)) == CSF_OK;
        }
        for (int64_t t = 0; t < n_ticks;) {                    // (launches of at most 2^16 ticks: a second or less each)
            const int k = (int)std::min<int64_t>(n_ticks - t, 65536);
            Dev dd = e->d;
            dd.snap = (snap && t + k == n_ticks) ? e->snap_dev : nullptr;
            launch_small_tick(dd, k, e->main);
            e->mid_synced = false;
            HIPCHK(e, hipGetLastError());
            e->d.tick += k;
            e->moves += k;
            e->small_ticks += k;
            t += k;
        }
        e->device_ahead = true;
        if (snapped) *snapped = snap;
        return CSF_OK;
    }
    for (int64_t t = 0; t < n_ticks; t++) {
        rc = enqueue_tick(e, n_ticks - t);
        if (rc) return rc;
    }
    if ((rc = chase_join(e))) return rc;                          // (every other entry point works on the main stream alone)
    if (e->cal_phase >= 1 && e->cal_phase <= 3) e->cal_phase = 0;  // (a measurement does not span calls)
    if (n_ticks > 0) e->device_ahead = true;
    return CSF_OK;
}

int csf_step(csf_engine *e, int64_t n_ticks) { return step_impl(e, n_ticks, false, nullptr); }

int csf_sync(csf_engine *e) {
    if (!e) return CSF_E_ARG;
    HIPCHK(e, hipSetDevice(e->device));
    {
        int rc = flush_pending(e);
        if (rc) return rc;
    }
    // (polling hipStreamQuery before sleeping in the runtime was tried again in round 5, with the group's stream lifetime put
    // right: no abort any more, and no gain - SocialForceIntersection.step() 37 / 54 us per tick at N = 3 / 1 024 against 32 / 48)
    HIPCHK(e, hipStreamSynchronize(e->main));
    HIPCHK(e, hipStreamSynchronize(e->comm));
    chase_cal_resolve(e, true);
    if (e->chase_ticks != e->chase_checked) {   // did a wait of the side-by-side tick give up? (csf_dev.h: CHASE_SPIN_LIMIT - it never has)
        e->chase_checked = e->chase_ticks;
        if (!e->bound_pin) HIPCHK(e, hipHostMalloc((void **)&e->bound_pin, 2 * (size_t)e->cap * sizeof(double), hipHostMallocDefault));
        HIPCHK(e, hipMemcpyAsync(e->bound_pin, e->chase_misc.p + 1, sizeof(unsigned), hipMemcpyDeviceToHost, e->main));
        HIPCHK(e, hipStreamSynchronize(e->main));
        unsigned gave_up = 0;
        std::memcpy(&gave_up, e->bound_pin, sizeof gave_up);
        if (gave_up != 0)
            return fail(e, CSF_E_DEVICE, "%u waits of the per-agent kernel beside the pair launch gave up: the states since are not a simulation (CSF_CHASE=0 takes the two launches in turn)", gave_up);
    }
    return CSF_OK;
}

int csf_chase_ticks(const csf_engine *e, int64_t *n_ticks) {
    if (!e || !n_ticks) return CSF_E_ARG;
    *n_ticks = e->chase_ticks;
    return CSF_OK;
}

int csf_chase_calibration(const csf_engine *e, int32_t *side_by_side, double us_per_tick[2]) {
    if (!e) return CSF_E_ARG;
    if (side_by_side) *side_by_side = e->knobs.chase >= 2 ? 1 : e->knobs.chase == 0 ? -1 : e->chase_state;
    if (us_per_tick) us_per_tick[0] = e->chase_cal_us[0], us_per_tick[1] = e->chase_cal_us[1];
    return CSF_OK;
}

int csf_calc_forces(csf_engine *e) {
    if (!e) return CSF_E_ARG;
    HIPCHK(e, hipSetDevice(e->device));
    int rc = upload_all(e);
    if (rc) return rc;
    if (e->order.empty()) return CSF_OK;
    rc = wait_gather(e);
    if (rc) return rc;
    rc = bounds_before_pair(e);
    if (rc) return rc;
    if ((e->world > 1 || e->nccl != nullptr || e->loopback) && e->d.recs_valid && e->ticks_since_rebin > 1) {
        launch_sorted_copy(e->d, e->main, e->xbuf_fresh);
        e->xbuf_fresh = false;
    }
    if ((rc = set_fov_band(e))) return rc;
    if (e->d.n_live > 1) launch_pair_all(e, e->d);
    bounds_after_pair(e, false);                     // the records do not move: recompute next time
    launch_road(e->d, e->main);
    launch_agent(e->d, PH_DEST | PH_COMBINE, e->main);
    HIPCHK(e, hipGetLastError());
    e->device_ahead = true;
    return CSF_OK;
}

int csf_apply_forces(csf_engine *e, const double *Fx, const double *Fy) {
    if (!e) return CSF_E_ARG;
    if (!Fx || !Fy) return fail(e, CSF_E_ARG, "csf_apply_forces: NULL force array");
    HIPCHK(e, hipSetDevice(e->device));
    if (e->world > 1) return fail(e, CSF_E_STATE, "csf_apply_forces is a single-device entry point");
    int rc = upload_all(e);
    if (rc) return rc;
    if (e->order.empty()) return CSF_OK;
    HIPCHK(e, hipStreamSynchronize(e->main));
    {   // Fx, Fy are in population order, the device array is indexed by slot
        std::vector<double> fx((size_t)e->d.n, 0.0), fy((size_t)e->d.n, 0.0);
        for (size_t i = 0; i < e->order.size(); i++) {
            fx[(size_t)e->order[i]] = Fx[i];
            fy[(size_t)e->order[i]] = Fy[i];
        }
        HIPCHK(e, hipMemcpy(e->F.p, fx.data(), fx.size() * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(e, hipMemcpy(e->F.p + e->cap, fy.data(), fy.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    launch_agent(e->d, PH_INTEGRATE, e->main);
    HIPCHK(e, hipGetLastError());
    e->mid_synced = false;
    e->bounds_fresh = false;
    if (!e->pair_since_move) e->moved_unbinned++;              // (csf_calc_forces before it has counted this tick already)
    e->pair_since_move = false;
    e->moves++;
    e->d.tick++;
    e->device_ahead = true;
    return CSF_OK;
}

int csf_replay_forces(csf_engine *e, int64_t n_ticks, const double *Fx, const double *Fy, const int32_t *lengths,
                      int32_t fix_speed, int32_t stride, double *states_out) {
    if (!e) return CSF_E_ARG;
    if (n_ticks < 0 || stride < 1 || (n_ticks > 0 && (!Fx || !Fy))) return fail(e, CSF_E_ARG, "csf_replay_forces: bad arguments");
    if (e->world > 1) return fail(e, CSF_E_STATE, "csf_replay_forces is a single-device entry point");
    HIPCHK(e, hipSetDevice(e->device));
    int rc = upload_all(e);
    if (rc) return rc;
    if ((rc = ensure_compact(e))) return rc;
    const int64_t n = e->d.n, cap = e->cap;
    if (n == 0 || n_ticks == 0) return CSF_OK;
    HIPCHK(e, hipStreamSynchronize(e->main));
    const int64_t chunk = std::min<int64_t>(n_ticks, 256);
    DevBuf<double> fbuf, hbuf;
    DevBuf<int32_t> lbuf;
    // (the per-agent kernel asks for the destination-force rows F[2], F[3] of its view unconditionally - a guarded load would be
    // a branch with its own wait; the view says how many rows it has, Dev::F_rows, and the kernel clamps the row: no padding)
    HIPCHK(e, fbuf.alloc((size_t)chunk * 2 * (size_t)cap));
    const int64_t n_samples = n_ticks / stride;
    if (states_out && n_samples > 0) HIPCHK(e, hbuf.alloc((size_t)n_samples * (size_t)n * (size_t)e->d.ns));
    if (lengths) {
        HIPCHK(e, lbuf.alloc((size_t)n));
        HIPCHK(e, hipMemcpy(lbuf.p, lengths, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    Dev dd = e->d;                       // a view of the engine with replay forces and a private history ring
    dd.F_rows = 2;                       // (the view holds Fx, Fy of one tick: the kernel clamps its row index - csf_dev.h)
    dd.replay_len = lengths ? lbuf.p : nullptr;
    dd.hist = (states_out && n_samples > 0) ? hbuf.p : nullptr;
    dd.hist_stride = stride;
    dd.hist_cap = (int32_t)std::max<int64_t>(n_samples, 1);
    std::vector<double> host((size_t)chunk * 2 * (size_t)cap, 0.0);
    for (int64_t t0 = 0; t0 < n_ticks; t0 += chunk) {
        const int64_t cnt = std::min<int64_t>(chunk, n_ticks - t0);
        for (int64_t t = 0; t < cnt; t++) {
            memcpy(&host[(size_t)(t * 2) * cap], Fx + (t0 + t) * n, (size_t)n * sizeof(double));
            memcpy(&host[(size_t)(t * 2 + 1) * cap], Fy + (t0 + t) * n, (size_t)n * sizeof(double));
        }
        HIPCHK(e, hipMemcpy(fbuf.p, host.data(), (size_t)cnt * 2 * cap * sizeof(double), hipMemcpyHostToDevice));
        for (int64_t t = 0; t < cnt; t++) {
            dd.F = fbuf.p + (size_t)(t * 2) * cap;   // PH_INTEGRATE alone reads only Fx = F[0][.], Fy = F[1][.]
            dd.replay_tick = t0 + t;
            dd.tick = t0 + t;                        // history sample index counts from the start of the replay
            launch_agent(dd, PH_INTEGRATE | (fix_speed ? PH_FIXSPEED : 0), e->main);
        }
        HIPCHK(e, hipGetLastError());
        HIPCHK(e, hipStreamSynchronize(e->main));
    }
    e->d.tick += n_ticks;
    e->moves += n_ticks;
    e->mid_synced = false;
    e->moved_unbinned += n_ticks;
    e->pair_since_move = false;
    if (fix_speed) {                         // (calibration.py:454-458 sets the speed to |F|: no clamp bounds the step)
        e->bound_stale = true;
        e->moved_unbinned = 1 << 20;         // (nor what the candidate lists and the circles allow for: re-bin before the next pair launch)
    }
    e->device_ahead = true;
    e->bounds_fresh = false;
    if (dd.hist)
        HIPCHK(e, hipMemcpy(states_out, hbuf.p, (size_t)n_samples * n * e->d.ns * sizeof(double), hipMemcpyDeviceToHost));
    fbuf.release();
    hbuf.release();
    lbuf.release();
    return CSF_OK;
}

int csf_dest_force(csf_engine *e, double *Fx, double *Fy) {
    if (!e) return CSF_E_ARG;
    if (!Fx || !Fy) return fail(e, CSF_E_ARG, "csf_dest_force: NULL output");
    HIPCHK(e, hipSetDevice(e->device));
    int rc = upload_all(e);
    if (rc) return rc;
    if (e->order.empty()) return CSF_OK;
    launch_agent(e->d, PH_DEST, e->main);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(e->main));
    e->device_ahead = true;
    if ((rc = read_rows(e, e->F.p + 2 * e->cap, 1, Fx, false))) return rc;
    return read_rows(e, e->F.p + 3 * e->cap, 1, Fy, false);
}

int csf_get_state(csf_engine *e, double *s_out, int32_t *dest_ptr, uint8_t *znav, int64_t *tick) {
    if (!e) return CSF_E_ARG;
    HIPCHK(e, hipSetDevice(e->device));
    int rc = upload_all(e);  // a never-stepped engine still answers from a consistent device copy
    if (rc) return rc;
    rc = csf_sync(e);
    if (rc) return rc;
    const int64_t n = (int64_t)e->order.size();
    const int ns = e->d.ns;
    if (tick) *tick = e->d.tick;
    if (n == 0) return CSF_OK;
    if (s_out && (rc = read_rows(e, e->s.p, ns, s_out, true))) return rc;
    if (dest_ptr && (rc = read_rows(e, e->ptr.p, 1, dest_ptr, false))) return rc;
    if (znav) {
        std::vector<uint8_t> z((size_t)n);
        if ((rc = read_rows(e, e->znav.p, 1, z.data(), false))) return rc;
        for (int64_t a = 0; a < n; a++) {
            znav[3 * a + 0] = (z[(size_t)a] & 3) == 0;
            znav[3 * a + 1] = (z[(size_t)a] & 3) == 1;
            znav[3 * a + 2] = (z[(size_t)a] & 3) == 2;
        }
    }
    return CSF_OK;
}

static int get_F(csf_engine *e, int comp, double *out) {
    if (!out) return CSF_OK;
    return read_rows(e, e->F.p + (size_t)comp * e->cap, 1, out, false);
}

static int snap_unpack(csf_engine *e, double *s_out, int32_t *dest_ptr, uint8_t *znav, double *Fx, double *Fy);

int csf_get_tick(csf_engine *e, double *s_out, int32_t *dest_ptr, uint8_t *znav, double *Fx, double *Fy, int64_t *tick) {
    if (!e) return CSF_E_ARG;
    HIPCHK(e, hipSetDevice(e->device));
    int rc = upload_all(e);  // a never-stepped engine still answers from a consistent device copy
    if (rc) return rc;
    const int64_t n = (int64_t)e->order.size();
    const int ns = e->d.ns;
    if (tick) *tick = e->d.tick;
    if (n == 0) return csf_sync(e);
    if ((rc = sync_order(e))) return rc;
    if ((rc = snap_reserve(e, (size_t)n * ((size_t)(ns + 2) * sizeof(double) + sizeof(int32_t) + 3)))) return rc;
    launch_snapshot(e->d, e->snap_dev, e->main);
    HIPCHK(e, hipGetLastError());
    rc = csf_sync(e);
    if (rc) return rc;
    return snap_unpack(e, s_out, dest_ptr, znav, Fx, Fy);
}

// the packed read-back in the mapped host buffer -> the caller's arrays
static int snap_unpack(csf_engine *e, double *s_out, int32_t *dest_ptr, uint8_t *znav, double *Fx, double *Fy) {
    const int64_t n = (int64_t)e->order.size();
    const int ns = e->d.ns;
    const double *S = (const double *)e->snap_host;
    const double *F = S + (size_t)n * ns;
    const int32_t *P = (const int32_t *)(F + 2 * (size_t)n);
    const uint8_t *Z = (const uint8_t *)(P + n);
    if (s_out) std::memcpy(s_out, S, (size_t)n * ns * sizeof(double));
    if (Fx) std::memcpy(Fx, F, (size_t)n * sizeof(double));
    if (Fy) std::memcpy(Fy, F + n, (size_t)n * sizeof(double));
    if (dest_ptr) std::memcpy(dest_ptr, P, (size_t)n * sizeof(int32_t));
    if (znav) std::memcpy(znav, Z, (size_t)n * 3);
    return CSF_OK;
}

// csf_step(e, n_ticks) + csf_get_tick in one call: what a caller that looks at every tick does (SocialForceIntersection.step(),
// intersection.py:866-896 - the host mirror refreshes vehicle.s, znav, force after each tick).  On the one-wave path the kernel
// packs the read-back itself: one launch and one wait per call.
int csf_step_get_tick(csf_engine *e, int64_t n_ticks, double *s_out, int32_t *dest_ptr, uint8_t *znav, double *Fx, double *Fy,
                      int64_t *tick) {
    bool snapped = false;
    int rc = step_impl(e, n_ticks, true, &snapped);
    if (rc) return rc;
    if (!snapped) return csf_get_tick(e, s_out, dest_ptr, znav, Fx, Fy, tick);
    if (tick) *tick = e->d.tick;
    if ((rc = csf_sync(e))) return rc;
    return snap_unpack(e, s_out, dest_ptr, znav, Fx, Fy);
}

int csf_get_forces(csf_engine *e, double *Fx, double *Fy) {
    if (!e) return CSF_E_ARG;
    int rc = csf_sync(e);
    if (rc) return rc;
    if (e->dirty && !e->device_ahead) {
        rc = upload_all(e);
        if (rc) return rc;
    }
    if (e->order.empty()) return CSF_OK;
    if ((rc = get_F(e, 0, Fx))) return rc;
    return get_F(e, 1, Fy);
}

int csf_get_force_parts(csf_engine *e, double *Fdest_x, double *Fdest_y, double *Frep_x, double *Frep_y) {
    if (!e) return CSF_E_ARG;
    int rc = csf_sync(e);
    if (rc) return rc;
    if (e->order.empty()) return CSF_OK;
    if ((rc = get_F(e, 2, Fdest_x))) return rc;
    if ((rc = get_F(e, 3, Fdest_y))) return rc;
    if ((rc = get_F(e, 4, Frep_x))) return rc;
    return get_F(e, 5, Frep_y);
}

int csf_status(csf_engine *e, uint32_t *per_agent_flags) {
    if (!e) return CSF_E_ARG;
    if (!per_agent_flags) return fail(e, CSF_E_ARG, "csf_status: NULL output");
    int rc = upload_all(e);
    if (rc) return rc;
    rc = csf_sync(e);
    if (rc) return rc;
    if (e->order.empty()) return CSF_OK;
    rc = read_rows(e, e->status.p, 1, per_agent_flags, false);
    for (size_t i = 0; rc == CSF_OK && i < e->order.size(); i++) per_agent_flags[i] &= ~CSF_ST_EDGE;   // (internal: csf_dev.h)
    return rc;
}

int csf_enable_history(csf_engine *e, int32_t stride, int32_t capacity) {
    if (!e) return CSF_E_ARG;
    if (stride < 1 || capacity < 1) return fail(e, CSF_E_ARG, "stride and capacity must be >= 1");
    HIPCHK(e, hipSetDevice(e->device));
    int rc = csf_sync(e);
    if (rc) return rc;
    if ((rc = ensure_compact(e))) return rc;              // the ring is indexed by road user: slots == population order
    HIPCHK(e, e->hist.alloc((size_t)capacity * (size_t)e->cap * (size_t)e->d.ns));
    e->d.hist = e->hist.p;
    e->d.hist_stride = stride;
    e->d.hist_cap = capacity;
    return CSF_OK;
}

int csf_get_history(csf_engine *e, int64_t first_sample, int64_t n_samples, double *out) {
    if (!e) return CSF_E_ARG;
    if (!e->d.hist) return fail(e, CSF_E_STATE, "history is not enabled");
    if (!out || first_sample < 0 || n_samples < 0) return fail(e, CSF_E_ARG, "csf_get_history: bad arguments");
    int rc = csf_sync(e);
    if (rc) return rc;
    const int64_t have = e->d.tick / e->d.hist_stride;
    if (first_sample + n_samples > have || have - first_sample > e->d.hist_cap)
        return fail(e, CSF_E_ARG, "samples [%lld, %lld) are not in the ring (have %lld, capacity %d)",
                    (long long)first_sample, (long long)(first_sample + n_samples), (long long)have, e->d.hist_cap);
    const size_t row = (size_t)e->d.n * (size_t)e->d.ns;
    for (int64_t k = 0; k < n_samples; k++) {
        int64_t slot = (first_sample + k) % e->d.hist_cap;
        HIPCHK(e, hipMemcpy(out + (size_t)k * row, e->hist.p + (size_t)slot * row, row * sizeof(double), hipMemcpyDeviceToHost));
    }
    return CSF_OK;
}

int csf_pair_force(csf_engine *e, const double *src, int64_t m, const double *x, const double *y,
                   const double *psi, int32_t apply_fov, double *Fx, double *Fy) {
    if (!e) return CSF_E_ARG;
    if (m < 0 || !src || (m > 0 && (!x || !y || !psi || !Fx || !Fy))) return fail(e, CSF_E_ARG, "csf_pair_force: bad arguments");
    if (m == 0) return CSF_OK;
    HIPCHK(e, hipSetDevice(e->device));
    const csf_params &p = e->d.p;
    std::vector<float4> hs((size_t)m), hr((size_t)m);
    std::vector<float2> h2((size_t)m), ho((size_t)m);
    double ev = 0.0;  // vehicle.py:1062-1064
    if (src[3] > 0.0) ev = std::min(std::pow(src[3] / p.v_max_riding[1], 0.1), 0.7);
    for (int64_t k = 0; k < m; k++) {
        hs[(size_t)k] = make_float4(0.f, 0.f, (float)std::cos(src[2]), (float)std::sin(src[2]));
        hr[(size_t)k] = make_float4((float)(x[k] - src[0]), (float)(y[k] - src[1]), (float)std::cos(psi[k]), (float)std::sin(psi[k]));
        h2[(size_t)k] = make_float2((float)ev, (float)(1.0 / std::sqrt(1.0 - ev * ev)));
    }
    if (e->kat4.n < (size_t)(2 * m)) HIPCHK(e, e->kat4.alloc((size_t)(2 * m)));
    if (e->kat2.n < (size_t)(2 * m)) HIPCHK(e, e->kat2.alloc((size_t)(2 * m)));
    HIPCHK(e, hipMemcpy(e->kat4.p, hs.data(), (size_t)m * sizeof(float4), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(e->kat4.p + m, hr.data(), (size_t)m * sizeof(float4), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(e->kat2.p, h2.data(), (size_t)m * sizeof(float2), hipMemcpyHostToDevice));
    launch_pair_kat(e->d, e->kat4.p, e->kat2.p, e->kat4.p + m, m, apply_fov, e->kat2.p + m, e->main);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(e->main));
    HIPCHK(e, hipMemcpy(ho.data(), e->kat2.p + m, (size_t)m * sizeof(float2), hipMemcpyDeviceToHost));
    for (int64_t k = 0; k < m; k++) {
        Fx[k] = ho[(size_t)k].x;
        Fy[k] = ho[(size_t)k].y;
    }
    return CSF_OK;
}

int csf_untracked(csf_engine *e, uint8_t *out) {
    if (!e) return CSF_E_ARG;
    if (!out) return fail(e, CSF_E_ARG, "csf_untracked: NULL output");
    HIPCHK(e, hipSetDevice(e->device));
    int rc = upload_all(e);
    if (rc) return rc;
    const int64_t n = (int64_t)e->order.size();
    if (n == 0) return CSF_OK;
    if (n > 46340) return fail(e, CSF_E_ARG, "csf_untracked: the n x n matrix is limited to n <= 46340");
    if ((rc = wait_gather(e))) return rc;
    if ((rc = sync_order(e))) return rc;
    HIPCHK(e, e->scratch_u8.reserve((size_t)(n * n)));         // (every byte of the matrix is written by the kernel)
    e->d.state_current = e->state_all_current ? 1 : 0;        // (then every pair is decided on the fp64 state, as the reference does)
    launch_untracked(e->d, e->scratch_u8.p, e->main);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(e->main));
    HIPCHK(e, hipMemcpy(out, e->scratch_u8.p, (size_t)(n * n), hipMemcpyDeviceToHost));
    return CSF_OK;
}

static int nav_kat(csf_engine *e, int64_t n, const int32_t *idx, int what, const int32_t *stop, double *vd, double *ddest) {
    if (n < 0 || (n > 0 && !idx)) return fail(e, CSF_E_ARG, "bad agent list");
    for (int64_t k = 0; k < n; k++)
        if (idx[k] < 0 || idx[k] >= (int64_t)e->order.size()) return fail(e, CSF_E_ARG, "agent index %d out of range", idx[k]);
    HIPCHK(e, hipSetDevice(e->device));
    int rc = upload_all(e);
    if (rc) return rc;
    if (n == 0) return CSF_OK;
    HIPCHK(e, hipStreamSynchronize(e->main));                  // (the scratch buffers may still be read by an earlier call)
    HIPCHK(e, e->scratch_i32.reserve((size_t)(2 * n)));
    HIPCHK(e, e->scratch_f64.reserve((size_t)(2 * n)));
    int32_t *di = e->scratch_i32.p, *ds = e->scratch_i32.p + n;
    double *out = e->scratch_f64.p;
    std::vector<int32_t> slots((size_t)n);
    for (int64_t k = 0; k < n; k++) slots[(size_t)k] = e->order[(size_t)idx[k]];
    HIPCHK(e, hipMemcpy(di, slots.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
    if (stop) HIPCHK(e, hipMemcpy(ds, stop, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
    launch_nav_kat(e->d, di, n, what, stop ? ds : nullptr, out, out + n, e->main);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(e->main));
    e->device_ahead = true;
    if (vd) HIPCHK(e, hipMemcpy(vd, out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    if (ddest) HIPCHK(e, hipMemcpy(ddest, out + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return CSF_OK;
}

int csf_update_destination(csf_engine *e, int64_t n, const int32_t *idx) {
    if (!e) return CSF_E_ARG;
    return nav_kat(e, n, idx, 1, nullptr, nullptr, nullptr);
}

int csf_update_nav_state(csf_engine *e, int64_t n, const int32_t *idx, const int32_t *stop, double *vd, double *ddest) {
    if (!e) return CSF_E_ARG;
    if (n > 0 && (!vd || !ddest)) return fail(e, CSF_E_ARG, "csf_update_nav_state: NULL output");
    return nav_kat(e, n, idx, 2, stop, vd, ddest);
}

int csf_set_dest_pointer(csf_engine *e, int64_t n, const int32_t *idx, const int32_t *ptr) {
    if (!e) return CSF_E_ARG;
    if (n < 0 || (n > 0 && (!idx || !ptr))) return fail(e, CSF_E_ARG, "csf_set_dest_pointer: bad arguments");
    for (int64_t k = 0; k < n; k++) {
        if (idx[k] < 0 || idx[k] >= (int64_t)e->order.size()) return fail(e, CSF_E_ARG, "agent index %d out of range", idx[k]);
        const int32_t rows = (int32_t)(e->h_q[(size_t)e->order[(size_t)idx[k]]].size() / 3);
        if (ptr[k] < 0 || ptr[k] >= rows) return fail(e, CSF_E_ARG, "destination pointer %d outside the queue of agent %d (%d rows)", ptr[k], idx[k], rows);
    }
    HIPCHK(e, hipSetDevice(e->device));
    int rc = prepare_mutation(e);
    if (rc) return rc;
    for (int64_t k = 0; k < n; k++) e->h_ptr[(size_t)e->order[(size_t)idx[k]]] = ptr[k];
    return CSF_OK;
}

int csf_set_script(csf_engine *e, int64_t n, const int32_t *agent, const int64_t *offsets, const double *rows) {
    if (!e) return CSF_E_ARG;
    if (n < 0 || (n > 0 && (!agent || !offsets))) return fail(e, CSF_E_ARG, "csf_set_script: bad arguments");
    for (int64_t k = 0; k < n; k++) {
        if (agent[k] < 0 || agent[k] >= (int64_t)e->order.size()) return fail(e, CSF_E_ARG, "agent index %d out of range", agent[k]);
        if (offsets[k + 1] < offsets[k]) return fail(e, CSF_E_ARG, "offsets must be non-decreasing");
        if (offsets[k + 1] > offsets[k] && !rows) return fail(e, CSF_E_ARG, "csf_set_script: NULL rows");
        if (offsets[k + 1] - offsets[k] > 2000000000) return fail(e, CSF_E_ARG, "a prescribed trajectory holds at most 2e9 states");
    }
    if (n == 0) return CSF_OK;
    HIPCHK(e, hipSetDevice(e->device));
    int rc = prepare_mutation(e);                              // (through the host mirror: scripts are set once, before the run)
    if (rc) return rc;
    for (int64_t k = 0; k < n; k++) {
        std::vector<double> &sc = e->h_script[(size_t)e->order[(size_t)agent[k]]];
        sc.assign(rows + 4 * offsets[k], rows + 4 * offsets[k + 1]);
    }
    return CSF_OK;
}

int csf_comm_unique_id(uint8_t id_out[CSF_UNIQUE_ID_BYTES]) {
    if (!id_out) return CSF_E_ARG;
    if (!g_rccl.load()) return fail(nullptr, CSF_E_COMM, "%s", g_rccl.err.c_str());
    static_assert(sizeof(ncclUniqueId) == CSF_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, CSF_E_COMM, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
    memcpy(id_out, &id, sizeof id);
    return CSF_OK;
}

int csf_comm_init(csf_engine *e, const uint8_t id[CSF_UNIQUE_ID_BYTES], int32_t rank, int32_t world) {
    if (!e) return CSF_E_ARG;
    if (world < 1 || rank < 0 || rank >= world) return fail(e, CSF_E_ARG, "bad rank %d / world %d", rank, world);
    if (world > 64) return fail(e, CSF_E_ARG, "world > 64 is not supported");
    if (e->nccl) return fail(e, CSF_E_STATE, "communicator already initialised");
    HIPCHK(e, hipSetDevice(e->device));
    int rc = download_all(e);
    if (rc) return rc;
    e->rank = rank;
    e->world = world;
    e->dirty = true;
    set_shard(e);
    if (world == 1 && !id) return CSF_OK;
    if (!id) return fail(e, CSF_E_ARG, "unique id is NULL");
    if (!g_rccl.load()) return fail(e, CSF_E_COMM, "%s", g_rccl.err.c_str());
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    NCCLCHK(e, g_rccl.CommInitRank(&e->nccl, world, uid, rank));
    return CSF_OK;
}

int csf_comm_init_loopback(csf_engine *const *engines, int32_t world) {
    if (!engines || world < 1 || world > 64) return CSF_E_ARG;
    for (int r = 0; r < world; r++) {
        csf_engine *e = engines[r];
        if (!e) return CSF_E_ARG;
        if (e->nccl || e->loopback || e->world > 1) return fail(e, CSF_E_STATE, "engine already belongs to a communicator");
        if (e->device != engines[0]->device || e->order.size() != engines[0]->order.size() || e->d.p.model != engines[0]->d.p.model)
            return fail(e, CSF_E_ARG, "loopback members hold the same population on the same device");
        for (int q = 0; q < r; q++)
            if (engines[q] == e) return fail(e, CSF_E_ARG, "engine listed twice");
    }
    HIPCHK(engines[0], hipSetDevice(engines[0]->device));
    for (int r = 0; r < world; r++) {
        csf_engine *e = engines[r];
        int rc = download_all(e);
        if (rc) return rc;
        e->rank = r;
        e->world = world;
        e->loopback = true;
        e->dirty = true;
        e->group.assign(engines, engines + world);
        if (r > 0) {                                  // one stream for the whole group: ticks and exchanges in order
            HIPCHK(e, hipStreamSynchronize(e->main));
            e->main_hold = engines[0]->main_hold;     // (the member's own stream goes with its last holder: now)
            e->main = e->main_hold->s;
        }
        set_shard(e);
    }
    return CSF_OK;
}

int csf_step_group(csf_engine *const *engines, int32_t world, int64_t n_ticks) {
    if (!engines || world < 1 || !engines[0]) return CSF_E_ARG;
    csf_engine *e0 = engines[0];
    if (n_ticks < 0) return fail(e0, CSF_E_ARG, "n_ticks must be >= 0");
    if (!e0->loopback || (int)e0->group.size() != world) return fail(e0, CSF_E_STATE, "not a loopback group of %d engines", world);
    for (int r = 0; r < world; r++)
        if (engines[r] != e0->group[(size_t)r]) return fail(e0, CSF_E_ARG, "members must be passed in rank order");
    HIPCHK(e0, hipSetDevice(e0->device));
    for (int r = 0; r < world; r++) {
        int rc = upload_all(engines[r]);
        if (rc) return rc;
    }
    for (int64_t t = 0; t < n_ticks; t++) {
        for (int r = 0; r < world; r++) {
            int rc = !e0->order.empty() ? enqueue_tick(engines[r]) : (engines[r]->d.tick++, CSF_OK);
            if (rc) return rc;
        }
        if (!e0->order.empty()) {
            int rc = loopback_exchange(engines, world);
            if (rc) return rc;
        }
    }
    if (n_ticks > 0 && !e0->order.empty())
        for (int r = 0; r < world; r++) engines[r]->device_ahead = true;
    return CSF_OK;
}

int csf_shard_range(const csf_engine *e, int64_t *lo, int64_t *hi) {
    if (!e) return CSF_E_ARG;
    if (lo) *lo = e->d.lo;
    if (hi) *hi = e->d.hi;
    return CSF_OK;
}

int csf_far_radius(const csf_engine *e, double *radius_m) {
    if (!e || !radius_m) return CSF_E_ARG;
    // (a population that has not been uploaded yet: the kernel the upload's re-binning will choose)
    const int32_t pv = e->dirty ? pair_variant_for(e, (int64_t)e->order.size()) : e->d.pair_variant;
    const bool binned = pv == 0 && e->d.p.model != CSF_BICYCLE && e->d.n >= BIN_MIN_AGENTS;
    // (computed here rather than read back: the kernel's copy is refreshed with the next upload of the population)
    *radius_m = binned ? (double)(float)far_radius(e->far_kappa, e->d.n, e->knobs.far_eps) : (double)INFINITY;  // no circles, no cull
    // (large populations: the radius in use since the last re-binning, from the sources a receiver can meet - rebin)
    if (binned && e->far_T > 0.0 && e->far_kappa > 0.0) *radius_m = (double)(float)(e->far_T / e->far_kappa);
    return CSF_OK;
}

int csf_profile_enable(csf_engine *e, int32_t on) {
    if (!e) return CSF_E_ARG;
    e->profile = on > 0 ? on : 0;
    if (e->profile) {                    // the event pool is created here, not inside the first sampled tick
        HIPCHK(e, hipSetDevice(e->device));
        return prof_make_pool(e);
    }
    return CSF_OK;
}

int csf_profile_kernels(csf_engine *e, double ms[4], int64_t launches[4]) {
    if (!e) return CSF_E_ARG;
    int rc = csf_sync(e);
    if (rc) return rc;
    while (e->prof_resolved < e->prof_issued)
        if ((rc = prof_resolve_one(e))) return rc;
    for (int k = 0; k < 4; k++) {
        if (ms) ms[k] = e->prof_ms[k];
        if (launches) launches[k] = e->prof_cnt[k];
    }
    e->last_gather_ms = e->prof_ms[3];
    for (int k = 0; k < 4; k++) e->prof_ms[k] = 0, e->prof_cnt[k] = 0;
    e->prof_ticks = 0;
    for (auto &v : e->prof_us) v.clear();
    return CSF_OK;
}

int csf_profile_read(csf_engine *e, double *pair_ms, double *agent_ms, int64_t *launches) {
    double ms[4];
    int64_t cnt[4];
    int rc = csf_profile_kernels(e, ms, cnt);
    if (rc) return rc;
    if (pair_ms) *pair_ms = ms[0];
    if (agent_ms) *agent_ms = cnt[2] > 0 ? ms[2] * (double)cnt[0] / (double)cnt[2] : 0.0;   // scaled to the pair kernel's count
    if (launches) *launches = cnt[0];
    return CSF_OK;
}

int csf_profile_samples_of(csf_engine *e, int32_t kernel, double *us, int64_t capacity, int64_t *n_samples) {
    if (!e || !n_samples || capacity < 0 || (capacity > 0 && !us) || kernel < 0 || kernel > 3)
        return e ? fail(e, CSF_E_ARG, "csf_profile_samples_of: bad arguments") : CSF_E_ARG;
    int rc = csf_sync(e);
    if (rc) return rc;
    while (e->prof_resolved < e->prof_issued)
        if ((rc = prof_resolve_one(e))) return rc;
    const std::vector<float> &v = e->prof_us[kernel];
    const int64_t n = std::min<int64_t>(capacity, (int64_t)v.size());
    for (int64_t k = 0; k < n; k++) us[k] = v[(size_t)k];
    *n_samples = n;
    return CSF_OK;
}

int csf_profile_samples(csf_engine *e, double *pair_us, int64_t capacity, int64_t *n_samples) {
    return csf_profile_samples_of(e, 0, pair_us, capacity, n_samples);
}

int csf_count_pairs(csf_engine *e, int64_t counts[4], const char **kernel_name) {
    if (!e || !counts) return e ? fail(e, CSF_E_ARG, "csf_count_pairs: NULL output") : CSF_E_ARG;
    HIPCHK(e, hipSetDevice(e->device));
    int rc = upload_all(e);
    if (rc) return rc;
    Dev &d = e->d;
    const char *name = pair_kernel_name(d);
    if (!e->segs.empty()) {                                      // one launch per parameter set: the kernel of the first
        Dev d0 = d;
        d0.p = e->classes[(size_t)e->segs[0].cls];
        d0.n_classes = 1;
        name = pair_kernel_name(d0);
    }
    if (kernel_name) *kernel_name = name;
    for (int k = 0; k < 4; k++) counts[k] = -1;
    if (std::string(name) != "pair_cull_kernel") return CSF_OK;   // only the cull-first kernel counts
    for (int k = 0; k < 4; k++) counts[k] = 0;
    if (d.n_live <= 1 || d.hi <= d.lo) return CSF_OK;
    if ((rc = wait_gather(e))) return rc;
    if (!e->segs.empty() && e->ticks_since_rebin >= e->knobs.rebin_ticks && (rc = rebin(e))) return rc;   // (arrivals since: flush_pending)
    if (d.classify && !e->bounds_fresh) launch_bounds(d, e->main);
    if ((e->world > 1 || e->nccl != nullptr || e->loopback) && d.recs_valid && e->ticks_since_rebin > 1) {
        launch_sorted_copy(d, e->main, e->xbuf_fresh);
        e->xbuf_fresh = false;
    }
    DevBuf<unsigned long long> &cnt = e->scratch_cnt;
    HIPCHK(e, cnt.reserve(4));
    HIPCHK(e, hipMemsetAsync(cnt.p, 0, 4 * sizeof(unsigned long long), e->main));
    if ((rc = set_fov_band(e))) return rc;
    Dev dd = d;                 // this tick's records and circles; the circles of the next tick are not touched
    dd.pair_count = cnt.p;
    dd.bnd_next = nullptr;
    dd.edge = nullptr;          // (no per-agent launch follows that would take undecided pairs over)
    launch_pair_all(e, dd);                                      // (the Bicycle-field launches of a mixed population do not count)
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(e->main));
    unsigned long long h[4] = {0, 0, 0, 0};
    HIPCHK(e, hipMemcpy(h, cnt.p, sizeof h, hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; k++) counts[k] = (int64_t)h[k];
    return CSF_OK;
}

int csf_near_dropped(csf_engine *e, int64_t *n_dropped) {
    if (!e || !n_dropped) return e ? fail(e, CSF_E_ARG, "csf_near_dropped: NULL output") : CSF_E_ARG;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->main));
    unsigned h = 0;
    HIPCHK(e, hipMemcpy(&h, e->edge_n.p + 1, sizeof h, hipMemcpyDeviceToHost));
    *n_dropped = (int64_t)h;
    return CSF_OK;
}

int csf_comm_stream_order(const csf_engine *e, int32_t *second_stream, double us_per_tick[2]) {
    if (!e || !second_stream) return CSF_E_ARG;
    *second_stream = e->comm_second ? 1 : 0;
    if (us_per_tick) us_per_tick[0] = e->comm_cal_us[0], us_per_tick[1] = e->comm_cal_us[1];
    return CSF_OK;
}

int csf_small_ticks(const csf_engine *e, int64_t *n_ticks) {
    if (!e || !n_ticks) return CSF_E_ARG;
    *n_ticks = e->small_ticks;
    return CSF_OK;
}

int csf_mid_ticks(const csf_engine *e, int64_t *n_ticks) {
    if (!e || !n_ticks) return CSF_E_ARG;
    *n_ticks = e->mid_ticks;
    return CSF_OK;
}

int csf_holes_taken(const csf_engine *e, int64_t *n) {
    if (!e || !n) return CSF_E_ARG;
    *n = e->holes.taken;
    return CSF_OK;
}

int csf_profile_gather(const csf_engine *e, double *gather_ms) {
    if (!e || !gather_ms) return CSF_E_ARG;
    *gather_ms = e->last_gather_ms;
    return CSF_OK;
}

}  // extern "C"
