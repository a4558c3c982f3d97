"""Index-sharding of one population over the GPUs of a node.

One process per GPU.  `torch.distributed` is used only for the rendezvous (broadcast of the RCCL unique
id) and for collecting results; the per-tick exchange — one all-gather of the fp32 source records
(x, y, cos psi, sin psi) over xGMI — is issued natively by the engine on its own HIP stream
(csf_comm_init / csf_step), so no Python runs inside the tick loop.

Every rank builds the same population (same arrays, same order); rank r integrates the receivers
[lo, hi) returned by `shard_bounds`.
"""
import numpy as np

SHARD_ALIGN = 64  # records per wave: shards are padded to whole waves (csf_engine.hip set_shard)


def shard_size(n, world):
    """Records per rank in the gathered buffer: ceil(n / world) rounded up to a multiple of 64."""
    per = -(-int(n) // int(world))
    return -(-per // SHARD_ALIGN) * SHARD_ALIGN


def shard_bounds(n, world, rank):
    """Receiver block [lo, hi) of `rank`; identical to csf_shard_range of the native engine."""
    if world <= 1:
        return 0, int(n)
    size = shard_size(n, world)
    lo = min(int(n), rank * size)
    return lo, min(int(n), lo + size)


def broadcast_unique_id(dist, rank, make_id, src=0):
    """Rank `src` creates the 128-byte RCCL unique id, every rank receives it."""
    box = [make_id() if rank == src else None]
    dist.broadcast_object_list(box, src=src)
    uid = box[0]
    if not isinstance(uid, (bytes, bytearray)) or len(uid) != 128:
        raise RuntimeError("unique id must be 128 bytes")
    return bytes(uid)


def shard_engine(engine, dist, rank, world):
    """Attach `engine` (already populated identically on every rank) to its shard."""
    if world <= 1:
        return shard_bounds(engine.n, 1, 0)
    uid = broadcast_unique_id(dist, rank, type(engine).comm_unique_id)
    engine.comm_init(uid, rank, world)
    lo, hi = engine.shard_range()
    if (lo, hi) != shard_bounds(engine.n, world, rank):
        raise RuntimeError(f"shard mismatch: engine {lo, hi} vs host {shard_bounds(engine.n, world, rank)}")
    return lo, hi


def gather_rows(dist, local_rows, n, world, rank):
    """All-gather per-agent rows: every rank contributes rows [lo, hi) and receives all n rows."""
    local_rows = np.ascontiguousarray(local_rows)
    lo, hi = shard_bounds(n, world, rank)
    if local_rows.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} must contribute {hi - lo} rows, got {local_rows.shape[0]}")
    if world <= 1:
        return local_rows
    parts = [None] * world
    dist.all_gather_object(parts, local_rows)
    out = np.concatenate(parts, axis=0)
    if out.shape[0] != n:
        raise RuntimeError("gathered row count does not match the population")
    return out


def gather_state(engine, dist, rank, world):
    """Full [n, n_states] state on every rank (each rank's own block is authoritative)."""
    lo, hi = shard_bounds(engine.n, world, rank)
    return gather_rows(dist, engine.state()[lo:hi], engine.n, world, rank)
