#!/usr/bin/env python3
"""bench.py — agent-steps/s of the stepping engine at N = 16,384 TwoDBicycle (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" is one population tick (SocialForceIntersection.step, intersection.py:866-896) of the
synthetic population of SURVEY.md §8(d): N agents uniform in a 200 m x 200 m square, random headings,
speeds in [3, 6] m/s, three destinations straight ahead.  For N > 1 the driver launches this file under
torch.distributed.run (one rank per GPU); the population is index-sharded, N stays 16,384 ("strong").
Rank 0 prints ONE JSON line.  The CPU oracle is used only for the `cpu_baseline` leg (N = 1, rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_TFLOPS = 157.3     # fp32 vector peak (same guide)
OPS_PER_PAIR = 100.0         # fp32 op-equivalents per pair evaluation


def synthetic_population(n, box, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, box, n)
    y = rng.uniform(0, box, n)
    psi = rng.uniform(-np.pi, np.pi, n)
    v = rng.uniform(3, 6, n)
    d = np.array([50.0, 99.0, 100.0])
    dq = np.zeros((n, 4, 3))
    dq[:, 0, 0] = x
    dq[:, 0, 1] = y
    dq[:, 1:, 0] = x[:, None] + d[None, :] * np.cos(psi)[:, None]
    dq[:, 1:, 1] = y[:, None] + d[None, :] * np.sin(psi)[:, None]
    s0 = np.c_[x, y, psi, v, np.zeros(n)]
    return s0, np.arange(n + 1) * 4, dq.reshape(-1, 3)


def measured_traffic():
    """HBM bytes per launch of the pair kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE are collected in their own runs, tools/pmc_passes.sh; bench.py cannot run under rocprofv3 itself)."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pair_kernel_pmc.json")))
    if not files:
        return None, None
    with open(files[-1]) as fh:
        rec = json.load(fh)
    return rec.get("hbm_bytes_per_launch"), os.path.relpath(files[-1], ROOT)


def cpu_baseline(n, box, ticks):
    """The oracle (CPU port of the reference algorithm) on this box's host cores, bounded sample."""
    from oracle import csf_oracle as orc

    s0, off, dq = synthetic_population(n, box)
    pop = orc.Population(orc.default_params("twod"), s0, 5.0, off, dq)
    pop.step(1)  # thread start-up, page faults
    t0 = time.perf_counter()
    pop.step(ticks)
    dt = time.perf_counter() - t0
    return {
        "value": n * ticks / dt, "unit": "agent-steps/s", "cores": orc.num_threads(), "kind": "port",
        "sample": f"{ticks} ticks of the same N={n} population, oracle/csf_oracle.c (fp64, OpenMP), {dt:.1f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--agents", type=int, default=16384)
    ap.add_argument("--box", type=float, default=200.0)
    ap.add_argument("--model", default="twod", choices=["twod", "bicycle", "invpend", "planarpoint"])
    ap.add_argument("--cpu-ticks", type=int, default=6, help="ticks of the CPU baseline sample (0 = skip)")
    ap.add_argument("--every-pair-steps", type=int, default=200,
                    help="ticks of the secondary run with the far-field cull switched off (0 = skip)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    rehearse = os.environ.get("CSF_BENCH_FORCE_DIST") == "1"  # 1-GPU rehearsal of the multi-rank code path
    if world > 1 or rehearse:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from cyclistsocialforce_amd import parameters
    from cyclistsocialforce_amd.engine import Engine
    from cyclistsocialforce_amd.parallel import shard_engine

    n, box = args.agents, args.box
    s0, off, dq = synthetic_population(n, box)
    if args.model == "invpend":
        s0 = np.c_[s0, np.zeros(n)]
    elif args.model == "planarpoint":
        s0 = s0[:, :4]
    eng = Engine(parameters.default_pod(args.model), n, device=local_rank)
    eng.add_agents(s0, 5.0)
    eng.set_dest_queue(np.arange(n), off, dq, reset=True)
    if world > 1:
        shard_engine(eng, dist, rank, world)
    elif rehearse:
        from cyclistsocialforce_amd.parallel import broadcast_unique_id
        eng.comm_init(broadcast_unique_id(dist, rank, Engine.comm_unique_id), rank, world)

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    eng.step(args.warmup)
    fence()
    eng.profile(8)   # HIP events around the pair kernel on every 8th tick of the timed region
    t0 = time.perf_counter()
    eng.step(args.steps)
    fence()
    dt = time.perf_counter() - t0
    pair_ms, agent_ms, launches = eng.profile_read()
    gather_ms = eng.profile_gather()
    eng.profile(0)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sanity: the timed run must have produced finite states and no status flags on this shard
    lo, hi = eng.shard_range()
    st = eng.state()[lo:hi]
    healthy = bool(np.isfinite(st).all() and (eng.status()[lo:hi] == 0).all())

    if rank == 0:
        value = n * args.steps / dt
        n_loc = hi - lo
        pair_s = pair_ms * 1e-3 / max(launches, 1)
        alg_bytes = 16.0 * n * n_loc + 8.0 * n_loc          # source records consumed + partial sums written
        pairs = float(n) * n_loc
        traffic, traffic_src = measured_traffic() if (world == 1 and n == 16384 and args.model == "twod") else (None, None)
        rfar = eng.far_radius()
        far_note = ("every pair evaluated" if not np.isfinite(rfar) else
                    f"batches of sources beyond {rfar:.1f} m skipped: together they add < 2^-24 f_0 to a receiver "
                    f"(DESIGN.md D8; CSF_FAR_EPS=0 evaluates every pair)")
        out = {
            "metric": "agent-steps/sec at N=16k TwoDBicycle", "value": value, "unit": "agent-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{n} {args.model} agents, uniform random in {box:g} m x {box:g} m, "
                                   f"all pairs, t_s=0.01", "agents": n, "rider_model": args.model,
                       "far_field": far_note,
                       "parallelism": f"index-sharded x{world}, RCCL all-gather of fp32 records per tick"
                       if world > 1 else "single GPU"},
            "healthy": healthy,
            "roofline": {
                "bound": "hbm", "kernel": "pair_kernel", "achieved": alg_bytes / pair_s / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": alg_bytes / pair_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": traffic_src, "algorithmic_bytes": alg_bytes,
                "launch_us": pair_s * 1e6, "agent_kernel_us": agent_ms * 1e3 / max(launches, 1),
                "all_gather_us": gather_ms * 1e3 / max(launches, 1),
                "note": "algorithmic bytes = 16 B x N sources per receiver (SURVEY.md 8(d)); served from LDS/L2, "
                        "so the kernel is VALU-bound: see valu.  Algorithmic = what the reference evaluates (every "
                        "pair); the kernel skips pairs outside the field of view and beyond the far-field radius",
            },
            "valu": {"achieved": OPS_PER_PAIR * pairs / pair_s / 1e12, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": OPS_PER_PAIR * pairs / pair_s / 1e12 / VALU_PEAK_TFLOPS,
                     "pairs_per_s": pairs / pair_s},
        }
        if world == 1 and not rehearse and np.isfinite(rfar) and args.every_pair_steps > 0:
            # the same population with the far-field cull off (reported beside the headline, never as `value`)
            os.environ["CSF_FAR_EPS"] = "0"
            ex = Engine(parameters.default_pod(args.model), n, device=local_rank)
            ex.add_agents(s0, 5.0)
            ex.set_dest_queue(np.arange(n), off, dq, reset=True)
            ex.step(40, sync=True)
            t0 = time.perf_counter()
            ex.step(args.every_pair_steps, sync=True)
            dte = time.perf_counter() - t0
            ex.close()
            del os.environ["CSF_FAR_EPS"]
            out["every_pair"] = {"value": n * args.every_pair_steps / dte, "unit": "agent-steps/s",
                                 "ms_per_step": dte / args.every_pair_steps * 1e3, "steps": args.every_pair_steps,
                                 "note": "CSF_FAR_EPS=0: no batch is skipped for distance"}
        if world == 1 and args.cpu_ticks > 0:
            out["cpu_baseline"] = cpu_baseline(n, box, args.cpu_ticks)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
