// csf_engine.hip - host side of the engine (the C ABI of include/csf.h): ONE translation unit, kept in sections under engine/ -
// the order below is the order of definition (internal helpers in an anonymous namespace, then the extern "C" entry points).

#include "engine/types.inc"   // includes, the RCCL entry points, device buffers, the knobs, struct csf_engine
#include "engine/consts.inc"   // errors, parameter checks, the far-field radius, the rounding bands of a tick, what the kernels derive from a parameter set
#include "engine/road_lattice.inc"   // the lattice over a large road network (csf_road.hip): cells, far-field fit
#include "engine/layout.inc"   // allocation, shard bounds, source chunks, the pair kernel by population size
#include "engine/binning.inc"   // holes left by departures, the re-binning, bounding circles around a pair launch
#include "engine/mirror.inc"   // host mirror <-> device: download, compaction, queue slab, parameter table, upload
#include "engine/gather.inc"   // a sharded run's blocks brought together before the population changes
#include "engine/pending.inc"   // population changes collected for the device (patch_kernel), order and row read-backs
#include "engine/comm_prof.inc"   // the tick's all-gather, per-kernel time stamps
#include "engine/abi_lifetime.inc"   // C ABI: version, errors, csf_create / csf_create_v / csf_destroy
#include "engine/abi_population.inc"   // C ABI: road users arriving and leaving, queues, roads, parameter sets, pushed states
#include "engine/tick.inc"   // one tick enqueued: two launches in turn, one launch (mid-size), side by side (large), one wave (a handful); csf_step, csf_sync
#include "engine/abi_forces_readback.inc"   // C ABI: forces on their own, replay, read-backs, the single-function entry points
#include "engine/abi_sharding.inc"   // C ABI: communicators, loopback groups
#include "engine/abi_measurement.inc"   // C ABI: far-field radius, time stamps, counters
