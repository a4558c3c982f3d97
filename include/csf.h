/*
 * csf.h — C ABI of the MI355X cyclist social-force stepping engine (libcsf_hip.so).
 *
 * The reference (chris-konrad/cyclistsocialforce) is pure Python and has no FFI of its own; the
 * drop-in seam is `SocialForceIntersection.step()` (intersection.py:866-896), i.e. one population
 * tick.  Each entry point below names the reference code it replaces.  The Python host
 * (cyclistsocialforce_amd/) binds these with ctypes; INTEGRATION.md shows the stub a maintainer of
 * the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; host arrays are borrowed for the duration of a call;
 *     outputs are written into caller-allocated buffers;
 *   - every function returns 0 on success or a negative csf_status_code; the message is available
 *     from csf_last_error(engine) (or csf_last_error(NULL) when creation itself failed);
 *   - an engine is bound to one HIP device and one host thread at a time (not re-entrant);
 *   - there is NO CPU fallback: without a usable gfx950 device csf_create fails with CSF_E_DEVICE.
 */
#ifndef CSF_H
#define CSF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSF_ABI_VERSION 9

/* rider models — vehicle.py:990 (Bicycle), :1292 (TwoDBicycle), :1651 (InvPendulumBicycle),
 * :1991 (PlanarPointBicycle), :2031 (PlanarBicycle); :920 (UncontrolledVehicle: follows a prescribed trajectory
 * - csf_set_script -, exerts the TwoDBicycle field with its own parameters, feels no force) */
enum csf_model { CSF_BICYCLE = 0, CSF_TWOD = 1, CSF_INVPEND = 2, CSF_PLANARPOINT = 3, CSF_PLANARBIKE = 4, CSF_UNCONTROLLED = 5,
                 CSF_BALANCINGRIDER = 6 /* vehicle.py:1953 (BalancingRiderBicycle: 8 states - x, y, psi, v, delta, phi, deltadot, phidot) */ };

/* priority rule — intersection.py:263, 739-741 */
enum csf_priority_rule { CSF_UNREGULATED = 0, CSF_P2R = 1 };

enum csf_status_code {
    CSF_OK = 0,
    CSF_E_ARG = -1,      /* bad argument (null pointer, size, index out of range) */
    CSF_E_DEVICE = -2,   /* no usable HIP device / HIP runtime error */
    CSF_E_CAPACITY = -3, /* more agents than the engine was created for */
    CSF_E_STATE = -4,    /* call not valid in the current engine state */
    CSF_E_COMM = -5,     /* RCCL error */
    CSF_E_ABI = -6,      /* the caller's csf_params (size) or ABI version is not this library's: csf_create_v */
    CSF_E_HOST = -7      /* the host ran out of memory (or another C++ exception was caught at the boundary: none crosses it);
                          * the engine is good for csf_last_error and csf_destroy only */
};

/* per-agent status bits reported by csf_status() (reference: exceptions / prints) */
#define CSF_ST_SPLINE 1u   /* vehicle.py:1495-1507: splprep would raise on duplicate points */
#define CSF_ST_NAN 2u      /* a force became non-finite (vehicle.py:1180-1185 "Isnan!") */
#define CSF_ST_NAVSTATE 4u /* vehicle.py:416-425: navigation state not one-hot */
#define CSF_ST_UNCONTROLLABLE 8u /* dynamics.py:1212-1214: PlanarBicycle at v <= 0 ("System not controllable!") */

/* POD mirror of the reference's parameter objects (parameters.py).  One set per engine = per vehicle
 * class; v_desired_default is per agent (csf_add_agents), whole parameter sets per agent: csf_set_param_classes.
 * Field order is ABI. */
typedef struct csf_params {
    /* VehicleParameters — parameters.py:430-508 */
    double t_s, d_arrived_inter, d_arrived_stop, v_max_stop, v_max_harddecel, hfov;
    double f_0, e_0, e_1, sigma_0, sigma_1, sigma_2, sigma_3;
    /* BicycleParameters — parameters.py:780-800 */
    double v_max_riding[2], p_decay, p_0, l, l_2, delta_max, a_max[2], a_desired_default[2];
    double k_p_v, k_p_delta, g;
    /* InvPendulumBicycleParameters — parameters.py:1429-1472 */
    double h, m, i_bike_longlong, i_steer_vertvert, c_steer, v_max_walk, delta_max_walk;
    /* PlanarPointBicycleParameters — parameters.py:1180-1201 (gain = -Re(pole), dynamics.py:933-940) */
    double k_psi;
    /* PlanarBicycleParameters — parameters.py:1203-1211: the two desired poles of the steer / yaw loop (re, im, re, im) */
    double pb_poles[4];
    /* BalancingRiderBicycleParameters — parameters.py:1214-1411; dynamics.py:261-705 (ABI 8).  The linearised Whipple-Carvallo
     * bicycle (Meijaard, Papadopoulos, Ruina & Schwab 2007: M q'' + v C1 q' + (g K0 + v^2 K2) q = f, q = (roll, steer)) as the
     * 2 x 2 blocks of its state matrix (row-major: M^-1 g K0, M^-1 K2, M^-1 C1), the steer-torque column of M^-1, the yaw row
     * (cos(lam) / w, cos(lam) c / w: dynamics.py:301-303), and the control model: the desired closed-loop poles as straight
     * lines over speed - (intercept, slope) of p0_real, p1_real, p1_imag, p2_real, p2_imag (parameters.py:1400-1409; fixed
     * poles: slopes 0) -, or fixed gains (br_mode 2: dynamics.py:604-605) */
    double br_minv_k0g[4], br_minv_k2[4], br_minv_c1[4], br_minv_steer[2], br_yaw[2], br_pole_fun[10], br_gains[5];
    int32_t model;         /* enum csf_model */
    int32_t priority_rule; /* enum csf_priority_rule */
    int32_t traj_len;      /* columns of the reference's traj ring buffer, int(30 / t_s) — vehicle.py:159 */
    int32_t br_mode;       /* BalancingRider: 0 poles from br_pole_fun at the current speed, 2 the gains br_gains */
} csf_params;

typedef struct csf_engine csf_engine;

/* ---- lifetime -------------------------------------------------------------------------------- */

/* SocialForceIntersection.__init__ (intersection.py:259-330): an empty population of one vehicle
 * class on HIP device `device`.  `n_capacity` bounds the number of agents.  Returns NULL on failure. */
csf_engine *csf_create(const csf_params *params, int64_t n_capacity, int32_t device);
/* The same for a caller that is NOT compiled against this header (ABI 9) - a ctypes / cffi binding declares csf_params by hand,
 * and a struct that has fallen behind the header (members were inserted in ABI 8) would make csf_create read `model` from
 * whatever lies behind the caller's memory.  csf_create_v refuses - NULL, csf_last_error(NULL) names both figures, nothing is
 * read from `params` - unless params_size == csf_params_size() and abi_version == csf_abi_version().  Bindings call this one;
 * csf_params_size() is what sizeof(csf_params) was when the library was built. */
csf_engine *csf_create_v(const csf_params *params, size_t params_size, int32_t abi_version, int64_t n_capacity, int32_t device);
size_t csf_params_size(void);
int csf_destroy(csf_engine *e);
const char *csf_last_error(const csf_engine *e);
int32_t csf_abi_version(void);

/* ---- population ------------------------------------------------------------------------------ */

/* Vehicle.__init__ (vehicle.py:64-204) for n agents at once: s0 is [n, n_states] row-major
 * (n_states = 5, 5, 6, 4, 5 by model), v_desired is [n].  Every new agent gets the single-row
 * destination queue (x0, y0, 0) of vehicle.py:183-185.  add_road_user: intersection.py:458-539. */
int csf_add_agents(csf_engine *e, int64_t n, const double *s0, const double *v_desired);

/* remove_road_user / remove_road_users_by_id (intersection.py:576-634): indices into the current
 * population, any order; the remaining agents keep their relative order. */
int csf_remove_agents(csf_engine *e, int64_t n, const int32_t *idx);

/* One traffic step in one call (ABI 9): the listed road users leave (as csf_remove_agents), n_arrive new ones join behind the remaining
 * ones (as csf_add_agents) and START with the given destination queues (CSR, at least one row each: what csf_set_dest_queue with
 * reset = 1 on the new road users would leave) - what SUMO co-simulation does every step (scenario.py:376-466: find_entered_exited,
 * remove_road_users_by_id, add_road_user + its route).  The same population and order as the three calls; one pass over the arrivals
 * instead of two, no start-row queue that is replaced at once. */
int csf_replace_agents(csf_engine *e, int64_t n_leave, const int32_t *idx_leave, int64_t n_arrive, const double *s0, const double *v_desired,
                       const int64_t *q_offsets, const double *q_rows);

/* How csf_add_agents / csf_remove_agents / csf_set_dest_queue reach the device once ticks have run (the per-tick arrivals
 * and departures of SUMO co-simulation, intersection.py:429-453, 458-634; scenario.py:376-466).  on != 0 (default): the
 * calls only record what to do, and the next device call applies the batch with one small launch - a removed road user
 * leaves a dead slot with a sentinel record, a new one takes a free slot whose place in the binned order lies in the tail
 * of sentinels behind the sorted batches, a replaced queue is appended to the queue slab (rewritten from the host's copy of
 * the queues when full); no download, no upload; the binned order is renewed when ticks x arrivals since the last renewal
 * reaches a few thousand.  The capacity holds up to 4096 extra slots for this.  on == 0: every change goes through the host
 * mirror (download, edit, upload, re-sort), which is also what sharded engines and engines with the history ring do.
 * The population order seen by every other entry point is the same either way.
 * On a sharded run (csf_comm_init) the population calls, csf_push_state and csf_set_integrator_state are COLLECTIVE: every rank
 * makes the same calls in the same order; before the first change after a tick the ranks exchange the fp64 state of their blocks
 * (a rank integrates only its own: one all-gather per state array on packed blocks), and the next tick starts from the upload with
 * new shard bounds.  The contract is the collective's: a call that only SOME ranks make blocks those ranks inside the exchange until
 * the others make it too - guard rank-dependent edits (e.g. a conditional push of vehicle.s) with the same condition on every rank. */
int csf_set_incremental(csf_engine *e, int32_t on);

/* Vehicle.setDestinations (vehicle.py:606-647) for n agents: CSR (offsets[n+1], xyz_stop[sum,3]);
 * reset = 0 appends to the agent's queue, reset = 1 replaces it and rewinds the pointer, reset = 2 replaces
 * it and keeps the pointer (rows edited in place: Vehicle.stop / Vehicle.go set the stop flag of the current
 * destination, vehicle.py:459-535). */
int csf_set_dest_queue(csf_engine *e, int64_t n, const int32_t *agent, const int64_t *offsets,
                       const double *xyz_stop, int32_t reset);

/* RoadEdge objects (intersection.py:214-242) flattened: CSR over vertices, one (F0, sigma) per edge. */
int csf_set_road_vertices(csf_engine *e, int32_t n_edges, const int64_t *offsets, const double *xy,
                          const double *F0, const double *sigma);

int csf_set_params(csf_engine *e, const csf_params *params);           /* parameter mutation between ticks */

/* Every reference vehicle owns its params object (vehicle.py:64-204): the field vehicle i exerts is evaluated with ITS
 * f_0 / sigma_* / e_* (vehicle.py:1592-1612; p_0 / p_decay for the Bicycle field, :1095-1101) and masked with ITS hfov
 * (intersection.py:733-735), and it steers, accelerates, brakes and arrives with its own gains and limits.
 * csf_set_param_classes installs 1 to 256 parameter sets (same t_s and traj_len as the engine; set 0 replaces the
 * engine's own, the priority rule stays the intersection's); csf_set_agent_class assigns sets to road users (new road
 * users start in set 0; a road user that has not taken a tick yet is re-initialised as the constructor of its set's
 * class would, vehicle.py:1728-1736).  The sets may be of different vehicle CLASSES (intersection.py:797-823 calls each
 * vehicle's own methods, so any mix may share an intersection): every row of s0 / csf_get_state then has the widest
 * layout among them (csf_num_states; states a class does not have stay 0), so install the sets before csf_add_agents.
 * With more than one set the engine evaluates the pair term either with a plain O(N^2) kernel that looks every source's
 * set up, or - from about 2048 road users per set, up to 16 sets, unsharded - with one launch of its culling kernel per
 * set over that set's run of the binned order.
 * An arrival whose set is assigned before the next device call (csf_add_agents, then csf_set_agent_class) takes it along in
 * its spawn record. */
int csf_set_param_classes(csf_engine *e, int32_t n_classes, const csf_params *classes);
int csf_set_agent_class(csf_engine *e, int64_t n, const int32_t *idx, const int32_t *cls);
int csf_set_v_desired(csf_engine *e, int64_t n, const int32_t *idx, const double *v_desired);
int csf_set_priority_rule(csf_engine *e, int32_t rule);                /* intersection.py:324 */

/* host-side mutation of vehicle.s between ticks (calibration.py:455-460): s is [n, n_states] */
int csf_push_state(csf_engine *e, int64_t n, const int32_t *idx, const double *s);

/* The hidden state of the rider models' integrators, which vehicle.s does not carry (ABI 7): x [n, 5] = vehicle.x of an
 * InvPendulumBicycle (vehicle.py:1728-1733, 1843: delta, ddelta, theta, dtheta, psi - the yaw UNWRAPPED, which is what the
 * commanded yaw arctan2(Fy, Fx) of vehicle.py:1832 is compared with), for PlanarBicycle x[., 0] = dynamics.x[0] (delta,
 * dynamics.py:195-197); psi_unwrapped [n] = the yaw state of the PlanarPoint / PlanarBicycle integrators (dynamics.py:828,
 * 943-966, 987-993); zrid [n, 2] = vehicle.zrid, one-hot riding / walking (vehicle.py:1735-1736, 1932-1950).  Any pointer
 * may be NULL.  csf_push_state replaces vehicle.s only (calibration.py:455-460 does the same to the reference object); a
 * caller that moves a population from one engine - or from the reference - to another needs these as well. */
int csf_get_integrator_state(csf_engine *e, double *x, double *psi_unwrapped, uint8_t *zrid);
int csf_set_integrator_state(csf_engine *e, int64_t n, const int32_t *idx, const double *x, const double *psi_unwrapped,
                             const uint8_t *zrid);

int64_t csf_num_agents(const csf_engine *e);
int32_t csf_num_states(const csf_engine *e);

/* ---- the hot call ---------------------------------------------------------------------------- */

/* n_ticks x SocialForceIntersection.step() (intersection.py:866-896): FOV mask + all-pairs repulsive
 * field + column sum + clamp (:690-845), destination force (vehicle.py:1416-1558), road edges
 * (:853-857), controller + kinematics (vehicle.py:1218-1272, 1810-1950; dynamics.py:996-1079) and the
 * position snapshot (:660-677).  Asynchronous: returns once the work is enqueued; csf_sync waits. */
int csf_step(csf_engine *e, int64_t n_ticks);
int csf_sync(csf_engine *e);

/* calc_forces() alone (intersection.py:747-864): forces of the next tick from the current snapshot.
 * Like the reference it advances the destination queue / navigation state of every agent. */
int csf_calc_forces(csf_engine *e);
/* vehicle.step(Fx, Fy) for every agent with caller-supplied forces (calibration.py:438-460 replay,
 * intersection.py:891-894).  Fx, Fy are [n]. */
int csf_apply_forces(csf_engine *e, const double *Fx, const double *Fy);
/* Calibration replay (calibration.py:438-460): n_ticks x vehicle.step(Fx[t][a], Fy[t][a]) for every agent with
 * recorded forces, in one call.  Fx, Fy are [n_ticks, n] row-major.  lengths is [n] or NULL: agent a is stepped
 * for its first lengths[a] ticks only (sequences of different length).  fix_speed != 0 sets v = |F| before every
 * step (calibration.py:454-458).  states_out is NULL or [n_ticks / stride, n, n_states]: the state after ticks
 * stride, 2*stride, ...  (vehicle.traj).  Single-device entry point. */
int csf_replay_forces(csf_engine *e, int64_t n_ticks, const double *Fx, const double *Fy, const int32_t *lengths,
                      int32_t fix_speed, int32_t stride, double *states_out);
/* calcDestinationForce() of every agent (vehicle.py:1189-1194, 1416-1558); mutates queue pointer and
 * navigation state exactly like the reference does. */
int csf_dest_force(csf_engine *e, double *Fx, double *Fy);

/* ---- read-back (all synchronise first) -------------------------------------------------------- */

/* s_out [n, n_states]; dest_ptr [n] (destpointer), znav [n,3] one-hot, tick = ticks since creation.
 * Any output pointer may be NULL. */
int csf_get_state(csf_engine *e, double *s_out, int32_t *dest_ptr, uint8_t *znav, int64_t *tick);
/* Everything the host mirror of SocialForceIntersection.step() refreshes after a tick (intersection.py:860-862,
 * 660-677; vehicle.py:1279-1282), in ONE device-to-host transfer: csf_get_state + csf_get_forces cost nine separate
 * copies of ~12 us each, which dominated the per-tick call at the reference's own population sizes (3 - 30
 * cyclists).  A kernel packs the row-major state, destination pointers, navigation state and total forces into a
 * pinned staging buffer; any output pointer may be NULL. */
int csf_get_tick(csf_engine *e, double *s_out, int32_t *dest_ptr, uint8_t *znav, double *Fx, double *Fy,
                 int64_t *tick);
/* total force of the last evaluated tick (vehicle.force, intersection.py:860-861) */
int csf_get_forces(csf_engine *e, double *Fx, double *Fy);
/* parts of it: destination force and clamped repulsive sum (before road edges) */
int csf_get_force_parts(csf_engine *e, double *Fdest_x, double *Fdest_y, double *Frep_x, double *Frep_y);
int csf_status(csf_engine *e, uint32_t *per_agent_flags);

/* opt-in history (vehicle.traj, vehicle.py:159, 1407-1410): record every `stride`-th tick into a
 * device ring of `capacity` samples; csf_get_history copies samples [first, first+n) (sample k = state
 * after tick (k+1)*stride) of all agents into out [n_samples, n_agents, n_states]. */
int csf_enable_history(csf_engine *e, int32_t stride, int32_t capacity);
int csf_get_history(csf_engine *e, int64_t first_sample, int64_t n_samples, double *out);

/* ---- single-function entry points for known-answer tests -------------------------------------- */

/* calcRepulsiveForce of one source at m receivers through the device code of the pair kernel
 * (vehicle.py:1560-1648 for TWOD/INVPEND/PLANARPOINT engines, :1054-1147 for BICYCLE engines).
 * src = (x, y, psi, v); receivers x, y, psi [m]; apply_fov != 0 also applies the mask of
 * intersection.py:690-745 (masked pairs return 0). */
int csf_pair_force(csf_engine *e, const double *src, int64_t m, const double *x, const double *y,
                   const double *psi, int32_t apply_fov, double *Fx, double *Fy);

/* get_untracked_foes() (intersection.py:690-745) as the matrix the reference returns: out [n, n] bytes, row = source i,
 * column = receiver j, 1 = "j ignores i" (the diagonal is 1), from the test the pair kernels apply.  n <= 46340. */
int csf_untracked(csf_engine *e, uint8_t *out);
/* Vehicle.updateDestination() (vehicle.py:545-594) for the listed agents, on its own: the queue pointer moves exactly as
 * it does at the start of calcDestinationForce(). */
int csf_update_destination(csf_engine *e, int64_t n, const int32_t *idx);
/* Vehicle.updateNavState(stop) (vehicle.py:354-457) for the listed agents: advances the three-state machine and returns
 * what the reference returns, (vd, ddest).  stop is NULL or [n]: >= 0 stands in for the stop flag of the current
 * destination (the reference passes it explicitly), < 0 reads the flag of the queue row. */
int csf_update_nav_state(csf_engine *e, int64_t n, const int32_t *idx, const int32_t *stop, double *vd, double *ddest);
/* vehicle.destpointer assigned from the host (Vehicle.stop types 1 and 2 step it back: vehicle.py:486-502) */
int csf_set_dest_pointer(csf_engine *e, int64_t n, const int32_t *idx, const int32_t *ptr);

/* UncontrolledVehicle.__init__(s0, trajectory) (vehicle.py:925-960): the prescribed trajectory of the listed road users
 * (of a CSF_UNCONTROLLED parameter set), rows (x, y, psi, v) = the columns of the reference's `traj`; at tick i the
 * state becomes row i while there is one (vehicle.py:964-979).  An empty range restores the default: the reference's
 * ring of zeros behind the start state (vehicle.py:158-160), i.e. the vehicle sits at (0, 0) from its first tick on. */
int csf_set_script(csf_engine *e, int64_t n, const int32_t *agent, const int64_t *offsets, const double *rows);

/* ---- sharding over the GPUs of one node ------------------------------------- */

/* The population is replicated on the host side of every rank; rank r integrates the contiguous
 * receiver block [r*N/world, (r+1)*N/world) and all-gathers the fp32 source records (x, y, cos psi,
 * sin psi [, e, k]) of the other blocks with one RCCL all-gather per tick on a second HIP stream.
 * Call csf_comm_unique_id on rank 0, distribute the 128 bytes out of band (torch.distributed), then
 * csf_comm_init on every rank before the first csf_step.  world == 1 needs none of this. */
#define CSF_UNIQUE_ID_BYTES 128
int csf_comm_unique_id(uint8_t id_out[CSF_UNIQUE_ID_BYTES]);
int csf_comm_init(csf_engine *e, const uint8_t id[CSF_UNIQUE_ID_BYTES], int32_t rank, int32_t world);
/* the receiver block of this rank: [lo, hi) */
int csf_shard_range(const csf_engine *e, int64_t *lo, int64_t *hi);

/* Rehearsal of the sharded path on ONE device (tests; no reference counterpart): `world` engines holding the same
 * population become ranks 0 .. world-1 of a loopback group.  They share one HIP stream, and where the ranks of a real
 * run call ncclAllGather the group copies every member's record block into the record arrays of the others.  Everything
 * else - shard bounds and padding, sentinel records, the receiver lists, the re-binning from gathered records, the
 * stale fp64 state of foreign agents - is the code path of csf_comm_init.  Members are stepped together with
 * csf_step_group (engines in rank order); csf_step on a member fails. */
int csf_comm_init_loopback(csf_engine *const *engines, int32_t world);
int csf_step_group(csf_engine *const *engines, int32_t world, int64_t n_ticks);

/* ---- measurement ------------------------------------------------------------------------------ */

/* HIP-event time of the pair kernel, measured on the stream it is launched on: enable with on = k > 0 to
 * bracket the kernel on every k-th tick (an event record costs a few microseconds of launch gap, so the timed
 * region of bench.py samples every 8th tick), run csf_step, then read the accumulated milliseconds and the
 * number of sampled launches (reset on read). */
int csf_profile_enable(csf_engine *e, int32_t on);
int csf_profile_read(csf_engine *e, double *pair_ms, double *agent_ms, int64_t *launches);
/* The same with every kernel of the tick: ms[4] = accumulated milliseconds of the pair kernel, the road-edge kernel, the
 * per-agent kernel (each from the start / end time stamps of the kernel's own dispatch) and the all-gather, launches[4]
 * = the number of launches behind each sum; resets the sums.  The pair kernel is timed on every sampled tick, the
 * others on every 8th of them (a time stamp costs a few microseconds of launch gap).  The events come from a fixed pool
 * that is recycled in order, so profiling may stay enabled for any number of ticks.  csf_profile_samples copies the
 * pair-kernel time of every sampled launch since the last reset (microseconds, at most `capacity`, at most 65 536 are
 * kept) WITHOUT resetting. */
int csf_profile_kernels(csf_engine *e, double ms[4], int64_t launches[4]);
int csf_profile_samples(csf_engine *e, double *pair_us, int64_t capacity, int64_t *n_samples);
/* ... of any of the four (ABI 9): kernel = 0 pair, 1 road, 2 per-agent, 3 all-gather - what a median needs (a mean of ~40 sampled
 * launches is moved by one launch that met a re-binning or a clock step; bench.py and tools/ report median, min and max). */
int csf_profile_samples_of(csf_engine *e, int32_t kernel, double *us, int64_t capacity, int64_t *n_samples);
/* What ONE launch of the pair kernel on the current snapshot does (the roofline of bench.py is computed from it):
 *   counts[0]  pair evaluations - calls of the force field vehicle.py:1560-1648 for a (source, receiver) pair, after the
 *              mask of intersection.py:690-745, the far-field cull and the per-pair reach test;
 *   counts[1]  sources put through the per-lane tests (field of view and reach);
 *   counts[2], counts[3]  full (128 pairs) and partial evaluation passes.
 * Runs one extra launch with device counters; the state of the simulation is not advanced.  All -1 when the engine's
 * pair kernel does not count (kernel_name, if not NULL, names the kernel either way). */
int csf_count_pairs(csf_engine *e, int64_t counts[4], const char **kernel_name);
/* Pairs the pair kernels could NOT hand to their exact path since the engine was created (0 in every run seen so far).
 * The kernels decide the field of view (intersection.py:690-745) in fp32 and note every pair inside the rounding band of
 * an edge, and every pair closer than 1 m, in a per-wave list of 32 entries (emptied between receivers when half full);
 * noted pairs are re-decided and re-evaluated from the precise records, and what is undecidable even there goes to the
 * per-agent kernel, which decides as the reference does (fp64 atan2 -> limitAngle -> angleDifference).  An entry that
 * found its list full, or a hand-over ring that overflowed within one tick, is counted here; such a pair keeps the fp32
 * result. */
int csf_near_dropped(csf_engine *e, int64_t *n_dropped);
/* Where a sharded engine issues the all-gather of a tick: 0 in stream order on its main stream, 1 on a second stream beside
 * the destination-force phase of the next tick.  Unless CSF_COMM_STREAM=main|second says which, the communicator times both
 * orders itself on its first tick - 32 launches each of the pair kernel + the collective on the records as they are, the
 * slowest rank's times shared through the communicator so that every rank decides alike - and keeps the faster;
 * us_per_tick (may be NULL) receives the two measurements (0 when nothing was measured). */
int csf_comm_stream_order(const csf_engine *e, int32_t *second_stream, double us_per_tick[2]);
/* milliseconds between the end of the agent kernel and the end of the RCCL all-gather, accumulated over the launches
 * of the last csf_profile_read (0 for an unsharded engine) */
int csf_profile_gather(const csf_engine *e, double *gather_ms);
/* Ticks this engine has run in its one-wave kernel (ABI 6).  Up to 32 road users of one parameter set (any rider class; not
 * UncontrolledVehicle) - the reference's own scenarios, e.g. the three cyclists of scenarios/ - on one device, with no road or a small one (at most 2048 vertices), without history ring or profiling: csf_step(e, n) is then ONE
 * launch of one wave for all n ticks (lane = road user; field-of-view decisions and np.sign(phi) on the fp64 difference of
 * the two positions, inside the band of fp32 rounding by the reference's own fp64 chain, intersection.py:690-745,
 * vehicle.py:1617-1625), instead of a pair launch and a per-agent launch per tick.  CSF_FUSED_SMALL=0 or a pinned CSF_PAIR_VARIANT keep the general path. */
int csf_small_ticks(const csf_engine *e, int64_t *n_ticks);
/* Ticks this engine has run as ONE launch each (ABI 7; csf_mid.hip).  Between 33 and ~3 000 road users of one parameter set on
 * one device - BASELINE config 2, and what the reference runs under SUMO - a tick is launch latency, not work: the pair sums
 * of a receiver group (intersection.py:690-745, 814-843) and the per-agent tick of its road users (:841-862, 891-892) share
 * one grid, the group's last workgroup to finish its sums carrying on with the road users; the next tick's records go to the
 * other half of a double buffer.  CSF_FUSED_MID=0, a pinned CSF_PAIR_VARIANT or per-kernel profiling keep the two launches. */
int csf_mid_ticks(const csf_engine *e, int64_t *n_ticks);
/* Ticks whose per-agent launch ran BESIDE the pair launch that feeds it (ABI 9).  From a few thousand road users
 * of one TwoD-field class on one device, a tick of csf_step(e, n >= 4) is a pair launch on one of the engine's two streams and the
 * per-agent kernel on the other: every wave runs its destination-force phase (vehicle.py:1416-1558) at once and takes up the column
 * sums of its 64 road users (intersection.py:841-862) as soon as the pair workgroups that form them have arrived, while the rest of the
 * pair launch drains; the next tick's records go to the other half of a double buffer.  Bit-identical to the two launches in turn.
 * Per-tick calls, re-binning ticks, roads, history, shards and several parameter sets take the launches in turn.
 * Whether it pays depends on the runtime giving the engine's two streams hardware queues of their own (two streams on one queue
 * serialise, and the tick is then a launch LONGER): unless CSF_CHASE=0 (never) or 2 (always) says otherwise, an engine times three whole periods
 * between re-binnings (64 ticks each: in turn, side by side, in turn) in its first call of 200 ticks or more after tick 256 (clocks that
 * have come up; the mean of the two in-turn periods cancels what is left of the ramp) and keeps the faster - engines of the same kind created later in the process take the finding over - csf_chase_calibration: side_by_side 1 / -1 / 0 (not measured yet),
 * us_per_tick = {in turn, side by side} (0 when nothing was measured).  Both ways give the same states. */
int csf_chase_ticks(const csf_engine *e, int64_t *n_ticks);
int csf_chase_calibration(const csf_engine *e, int32_t *side_by_side, double us_per_tick[2]);

/* Arrivals that took the slot of a road user who had left from nearby (ABI 8).  Under traffic - road users arriving and leaving
 * every tick, intersection.py:458-634 - a slot freed inside a batch of the binned order is handed to the next arrival that starts
 * within that batch's bounding circle (up to CSF_HOLE_DIST median radii from its centre, [1.0]; CSF_HOLE_REUSE=0 switches it off):
 * the circle does not grow and the sentinel tail of the order, which every receiver tests source by source, fills more slowly. */
int csf_holes_taken(const csf_engine *e, int64_t *n);
/* csf_step(e, n_ticks) followed by csf_get_tick(...) in one call (ABI 6): what a caller that looks at every tick does -
 * SocialForceIntersection.step() refreshes vehicle.s, znav and force after each tick (intersection.py:866-896).  On the
 * one-wave path the kernel packs the read-back itself behind its last tick: one launch and one wait per call.  Arguments as
 * csf_get_tick's (any output may be NULL). */
int csf_step_get_tick(csf_engine *e, int64_t n_ticks, double *s_out, int32_t *dest_ptr, uint8_t *znav, double *Fx, double *Fy,
                      int64_t *tick);

/* Far-field radius of the pair kernel (metres; +inf when the cull is off).  The repulsive field of
 * vehicle.py:1560-1648 decays at least like f_0 exp(-kappa rho); sources beyond
 * R = ln(n / eps) / kappa together add less than eps * f_0 (eps = 2^-24 unless the environment variable
 * CSF_FAR_EPS says otherwise; CSF_FAR_EPS=0 evaluates every pair) to a receiver's force, which is below
 * the resolution of the fp32 column sum; batches of binned source records entirely beyond R are skipped.
 * The reference has no such cut-off: this is the engine's only approximation besides fp32. */
int csf_far_radius(const csf_engine *e, double *radius_m);

#ifdef __cplusplus
}
#endif
#endif /* CSF_H */
