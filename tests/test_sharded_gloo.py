"""world_size = 2 rehearsal of the sharded tick on CPU (gloo).

The product's per-tick exchange is a native RCCL all-gather inside csf_step and needs GPUs; what runs here is
the host logic that surrounds it (cyclistsocialforce_amd.parallel: shard bounds, unique-id broadcast,
result gathering) with the CPU oracle standing in for the kernels, so that the decomposition itself —
receivers sharded by index, sources all-gathered as (x, y, psi, v) every tick — is checked against an
unsharded run: it must be bit-identical."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _population(n, box, seed=3):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, box, n); y = rng.uniform(0, box, n)
    psi = rng.uniform(-np.pi, np.pi, n); v = rng.uniform(3, 6, n)
    d = np.array([20.0, 39.0, 40.0])
    dq = np.zeros((n, 4, 3))
    dq[:, 0, 0] = x; dq[:, 0, 1] = y
    dq[:, 1:, 0] = x[:, None] + d * np.cos(psi)[:, None]
    dq[:, 1:, 1] = y[:, None] + d * np.sin(psi)[:, None]
    return np.c_[x, y, psi, v, np.zeros(n)], np.arange(n + 1) * 4, dq.reshape(-1, 3)


def _worker(rank, world, port, n, ticks, model, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    import torch.distributed as dist

    from cyclistsocialforce_amd import parallel
    from oracle import csf_oracle as orc

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        uid = parallel.broadcast_unique_id(dist, rank, lambda: bytes(range(128)))
        assert uid == bytes(range(128))
        s0, off, dq = _population(n, 30.0)
        pop = orc.Population(orc.default_params(model), s0, 5.0, off, dq)
        lo, hi = parallel.shard_bounds(n, world, rank)
        for _ in range(ticks):
            pop.calc_forces_range(lo, hi)
            pop.integrate_range(lo, hi)
            pop.update_snapshot_range(lo, hi)
            snap = parallel.gather_rows(dist, pop.snapshot(lo, hi), n, world, rank)   # the per-tick all-gather
            pop.set_snapshot(0, n, snap)
        full = parallel.gather_rows(dist, pop.state()[lo:hi], n, world, rank)
        if rank == 0:
            np.save(os.path.join(out_dir, "sharded.npy"), full)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("model,n", [("twod", 150), ("bicycle", 130)])
def test_two_rank_sharded_tick_equals_unsharded(tmp_path, model, n):
    import torch.multiprocessing as mp

    from oracle import csf_oracle as orc

    ticks = 12
    mp.spawn(_worker, args=(2, _free_port(), n, ticks, model, str(tmp_path)), nprocs=2, join=True)
    sharded = np.load(tmp_path / "sharded.npy")
    s0, off, dq = _population(n, 30.0)
    pop = orc.Population(orc.default_params(model), s0, 5.0, off, dq)
    pop.step(ticks)
    np.testing.assert_array_equal(sharded, pop.state())
