#!/usr/bin/env python3
"""bench.py — agent-steps/s of the stepping engine at N = 16,384 TwoDBicycle (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" is one population tick (SocialForceIntersection.step, intersection.py:866-896) of the
synthetic population of SURVEY.md §8(d): N agents uniform in a 200 m x 200 m square, random headings,
speeds in [3, 6] m/s, three destinations straight ahead.  For N > 1 the driver launches this file under
torch.distributed.run (one rank per GPU); the population is index-sharded, N stays 16,384 ("strong").
Rank 0 prints ONE JSON line.  The CPU oracle is used only for the `cpu_baseline` leg (N = 1, rank 0).

What the line reports beside `value`:
  roofline      the dominant kernel (the all-pairs kernel) against the fp32 vector peak: 100 op-equivalents per pair
                 x the pairs the launch actually EVALUATES (counted on the device, csf_count_pairs:
                pairs masked by the field of view are evaluated neither here nor by the reference) / the kernel's MEDIAN
                duration from HIP events on its own stream over the timed region.  `traffic` is the HBM bytes per launch
                from the committed rocprofv3 --pmc passes of this kernel, `hbm_frac` the physical HBM fraction.
  algorithmic   SURVEY.md §8(d)'s byte/flop rates over ALL N x n_loc pairs (what BASELINE.json's "HBM GB/s fraction" is
                quoted on).  These are rates of a stream that is served from LDS, not fractions of a peak.
  preroll_ticks untimed ticks of a SCRATCH engine (same population, thrown away) that bring the GPU out of its idle power
                state before anything is timed: a fresh process finds the first ticks up to 1.7 x slower
                (profiles/r2_tick_sequence.json).  The W warm-up ticks run on it too, but for the first min(W, 8), which the
                timed engine runs itself (first launches, first re-binning): the timed window starts at tick <= 8 of the
                population `config.workload` names (`timed_window.start`).
  dispersed     the same engine from tick ~740 on, when the crowd has spread to ~270 m (what rounds 1 - 3 timed).
"""
import argparse
import glob
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_TFLOPS = 157.3     # fp32 vector peak (same guide): 256 CUs x 4 SIMDs x 32 lanes x 2 flop x 2.4 GHz
OPS_PER_PAIR = 100.0         # fp32 op-equivalents per pair evaluation
OPS_PER_TEST = 25.0          # ... per source put through the per-lane field-of-view / reach test: 15 packed-lane operations,
                             # one rsq (quarter rate: x 4), the compare and the queue append (csf_pair.hip: keep_x2, sift2)


def synthetic_population(n, box, seed=0, reach=(50.0, 99.0, 100.0)):
    """uniform positions and headings, speeds in [3, 6] m/s, destinations straight ahead at `reach`
    metres (the default three of §8(d) last ~2 000 ticks at 5 m/s; long runs pass more, tools/large_configs.py)."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, box, n)
    y = rng.uniform(0, box, n)
    psi = rng.uniform(-np.pi, np.pi, n)
    v = rng.uniform(3, 6, n)
    d = np.asarray(reach, dtype=float)
    k = d.size + 1
    dq = np.zeros((n, k, 3))
    dq[:, 0, 0] = x
    dq[:, 0, 1] = y
    dq[:, 1:, 0] = x[:, None] + d[None, :] * np.cos(psi)[:, None]
    dq[:, 1:, 1] = y[:, None] + d[None, :] * np.sin(psi)[:, None]
    s0 = np.c_[x, y, psi, v, np.zeros(n)]
    return s0, np.arange(n + 1) * k, dq.reshape(-1, 3)


def tiled_curve_road(box, pitch=100.0):
    """scenarios/curve-scenario.py:63-81 geometry repeated on a `pitch` grid:
    (offsets, vertices, F0, sigma) as csf_set_road_vertices takes them."""
    from cyclistsocialforce_amd import parameters
    from cyclistsocialforce_amd.intersection import (CurvedRoadSegment, RoadSegmentCollection, StraightRoadSegment,
                                                     flatten_road_elements)

    rp = parameters.RoadElementParameters(sigma=2.0, F_0=0.15)
    els = []
    for gx in np.arange(0, box, pitch):
        for gy in np.arange(0, box, pitch):
            s1 = StraightRoadSegment(np.array((gx + 10.0, gy + 5.0, np.pi / 2)), 5, 25, params=rp, ds=0.1)
            s2 = CurvedRoadSegment(s1.x1, 5, 10, np.pi / 2, "right", params=rp, ds=0.1)
            s3 = CurvedRoadSegment(s2.x1, 5, 10, np.pi / 2, "left", params=rp, ds=0.1)
            s4 = StraightRoadSegment(s3.x1, 5, 20, params=rp, ds=0.1)
            els.append(RoadSegmentCollection((s1, s2, s3, s4)))
    return flatten_road_elements(els)


def build_id():
    """what identifies the build of libcsf_hip.so that is loaded (the first 16 hex digits of its SHA-256)"""
    import hashlib

    from cyclistsocialforce_amd import _ffi

    with open(_ffi.LIB_PATH, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE are
    collected in their own runs, tools/pmc_passes.sh; bench.py cannot run under rocprofv3 itself).  The newest file by
    (round, version) whose recorded kernel name matches is taken - and refused when it was measured on another build of
    the library than the one that is benched (`build_id`)."""
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", "r*_v*_pair_kernel_pmc.json")):
        m = re.match(r"r(\d+)_v(\d+)_", os.path.basename(path))
        if not m:
            continue
        with open(path) as fh:
            rec = json.load(fh)
        if kernel.split("<")[0] not in rec.get("kernel", ""):
            continue
        key = (int(m.group(1)), int(m.group(2)))
        if best is None or key > best[0]:
            best = (key, rec, path)
    if best is None:
        return None, None
    src = os.path.relpath(best[2], ROOT)
    if best[1].get("build_id") != build_id():
        return None, f"{src} was measured on build {best[1].get('build_id')}, this is build {build_id()}: not used"
    return best[1].get("hbm_bytes_per_launch"), src


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
        # the literal reference itself, which cannot run this N (memory ~ N^4; NaN beyond 101.6 m) - the
        # figures of BASELINE.md §2, measured in the survey container, not on this box (the reference never travels here)
        "literal_reference": {"kind": "literal-reference, survey container, 1 core (BASELINE.md §2; not measured in this run)",
                              "unit": "agent-steps/s", "twod_random": {"4": 1.3e3, "16": 1.0e3, "32": 5.7e2, "64": 57, "128": 2},
                              "demo_3_twod_700_ticks": 3.1e3},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--agents", type=int, default=16384)
    ap.add_argument("--box", type=float, default=200.0)
    ap.add_argument("--model", default="twod", choices=["twod", "bicycle", "invpend", "planarpoint"])
    ap.add_argument("--road", default="none", choices=["none", "curve-tiles"],
                    help="static-obstacle infrastructure (BASELINE config 5): the curve scenario tiled on a 100 m grid")
    ap.add_argument("--preroll", type=int, default=-1,
                    help="untimed ticks before the warm-up (reported as preroll_ticks); default: about 0.1 s of them")
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
    real_stdout = None
    if world > 1 or rehearse:
        import torch.distributed as dist

        # RCCL prints a version banner on stdout when a communicator is created: everything but the one JSON line goes
        # to stderr (the C-level descriptor is redirected, and restored right before the line is printed)
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse and world == 1:                              # (a plain `python bench.py`: no launcher has set these)
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("MASTER_PORT", "29533")
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
    road = tiled_curve_road(box) if args.road == "curve-tiles" else None

    def populate():
        e = Engine(parameters.default_pod(args.model), n, device=local_rank)
        e.add_agents(s0, 5.0)
        e.set_dest_queue(np.arange(n), off, dq, reset=True)
        if road is not None:
            e.set_road(*road)
        return e

    # Clocks first: a fresh process finds the GPU in a low power state, and the first ~200 ticks run up to 1.7 x slower
    # (profiles/r2_tick_sequence.json).  A SCRATCH engine (the same population, unsharded, on this rank's device) runs the
    # untimed pre-roll - about 0.1 s of ticks - and the W warm-up ticks, and is thrown away; the population the line names
    # is then created afresh and timed from its first ticks (`timed_window.start.tick`).
    if os.environ.get("CSF_BENCH_SWAP") == "1":                   # (measurement aid: which engine is created first)
        eng = populate()
        scratch = populate()
    else:
        scratch = populate()
        eng = populate()
    if world > 1:
        shard_engine(eng, dist, rank, world)
    elif rehearse:
        from cyclistsocialforce_amd.parallel import broadcast_unique_id
        eng.comm_init(broadcast_unique_id(dist, rank, Engine.comm_unique_id), rank, world)
    first = min(8, max(0, args.warmup))                       # untimed ticks of the timed engine itself (re-binning, first launches)

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def window_state(e_):
        """what the population looks like at an end of the timed window: extent (it disperses as it runs) and the work of
        one pair launch on it (device counters, one extra launch - outside the timed region)"""
        s_ = e_.state()
        w_, _ = e_.count_pairs(detail=True)
        return {"tick": int(e_.tick), "extent_m": [float(np.ptp(s_[:, 0])), float(np.ptp(s_[:, 1]))],
                "rms_radius_m": float(np.sqrt(((s_[:, :2] - s_[:, :2].mean(axis=0)) ** 2).sum(axis=1).mean())),
                "pairs_evaluated": None if w_ is None else int(w_["evaluated"]), "sources_tested": None if w_ is None else int(w_["tested"])}

    # the kernels' own start / end time stamps (hipExtLaunchKernelGGL events on the engine's stream).  A launch that carries
    # events costs the tick ~5 us (profiles/r4_v5_tick_sequence.json: 120.1 us per tick with every launch sampled, 114.6 with
    # none), so not every launch is sampled: every 16th tick in long runs (62 launches of the default 1 000 steps), every 4th tick
    # of the driver's 20-step command (five launches: their spread is 2 %).  (The event pool is created here, not in front of
    # the timed region.)
    every = max(1, min(16, args.steps // 5))
    if os.environ.get("CSF_BENCH_SAMPLE_EVERY"):              # (measurement aid: what do the time stamps themselves cost?)
        every = int(os.environ["CSF_BENCH_SAMPLE_EVERY"])
    eng.profile(every)
    eng.step(first)
    fence()
    scratch.step(first, sync=True)
    # (the start of the window is read off the scratch engine, which is in the very state the timed engine is in: a read-back
    # right before the timed region lets the device fall idle for a millisecond, and the pair kernel then runs ~10 % slower
    # for the next 100 - 200 ticks)
    win0 = window_state(scratch) if world == 1 else None
    preroll = args.preroll
    if preroll < 0:
        t0 = time.perf_counter()
        scratch.step(4, sync=True)
        tick_s = (time.perf_counter() - t0) / 4
        preroll = int(min(1000, max(0, 0.1 / max(tick_s, 1e-6))))
    eng.profile_kernels()                                     # (forget the time stamps of the first ticks)
    scratch.step(preroll + max(0, args.warmup - first), sync=True)
    # the scratch engine has dispersed by now (tick ~900: what rounds 1 - 3 reported as the headline): K steps of it, timed
    # the same way, go into the line as `dispersed` - and keep the device busy right up to the timed region
    dispersed = None
    if world == 1 and not rehearse:
        d0 = int(scratch.tick)
        t0 = time.perf_counter()
        scratch.step(args.steps, sync=True)
        dtd = time.perf_counter() - t0
        dispersed = {"value": n * args.steps / dtd, "unit": "agent-steps/s", "ms_per_step": dtd / args.steps * 1e3, "start_tick": d0,
                     "note": "the same population later on, when the crowd has spread to ~270 m (the window rounds 1 - 3 timed)"}
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step(args.steps)
    fence()
    dt = time.perf_counter() - t0
    samples = eng.profile_samples()
    stats = eng.profile_stats()                                # (median / min / max per kernel: before the reset below)
    prof = eng.profile_kernels()
    eng.profile(0)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sanity: the timed run must have produced finite states and no status flags on this shard
    lo, hi = eng.shard_range()
    st = eng.state()[lo:hi]
    healthy = bool(np.isfinite(st).all() and (eng.status()[lo:hi] == 0).all())
    work, kernel = eng.count_pairs(detail=True)     # one extra launch on the final snapshot, outside the timed region
    evaluated = None if work is None else work["evaluated"]
    win1 = window_state(eng) if world == 1 else None
    # the per-agent launch beside the pair launch (include/csf.h: csf_chase_ticks): what the scratch engine measured on its warm-up
    # ticks (16 ticks each way), and how many ticks of the timed engine took that order
    cal = scratch.chase_calibration()
    side_by_side = {"decided": {1: "side by side", -1: "in turn", 0: "not measured"}[cal[0]], "measured_us_per_tick": {"in_turn": cal[1][0], "side_by_side": cal[1][1]},
                    "timed_engine": {"decided": eng.chase_calibration()[0], "ticks_side_by_side": int(eng.chase_ticks()), "ticks": int(eng.tick)}}
    scratch.close()
    # every rank's own figures (kernel times incl. the all-gather, the stream order its communicator chose, the pairs one launch
    # of its receiver block evaluates): a scaling curve then explains itself
    order, cal = eng.comm_stream_order()
    med = lambda k: (stats[k]["median"] if stats.get(k) else 0.0)   # noqa: E731  (microseconds; 0: not launched / not sampled)
    mine = {"rank": rank, "receivers": [int(lo), int(hi)],
            "kernels_us": {k: med(k) for k in prof}, "kernels_us_stats": stats,
            "pairs_evaluated": evaluated, "comm_stream": order, "comm_calibration_us": cal}
    ranks = [mine]
    if dist is not None:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
    if rank == 0:
        value = n * args.steps / dt
        n_loc = hi - lo
        # MEDIAN duration of each kernel over its sampled launches (round 6; the means stay beside them in `kernels_us_stats`:
        # one launch of ~40 that meets a clock step or a re-binning's tail moves a mean by per cents, a median not at all)
        mean_s = {k: ms * 1e-3 / max(c, 1) for k, (ms, c) in prof.items()}
        med_s = {k: med(k) * 1e-6 for k in prof}
        pair_s, road_s, agent_s, launches = med_s["pair"], med_s["road"], med_s["agent"], prof["pair"][1]
        pairs_all = float(n) * n_loc
        alg_bytes = 16.0 * n * n_loc + 8.0 * n_loc          # source records consumed + partial sums written
        traffic, traffic_src = measured_traffic(kernel) if (world == 1 and n == 16384 and args.model == "twod") else (None, None)
        rfar = eng.far_radius()
        far_note = ("every pair evaluated" if not np.isfinite(rfar) else
                    f"batches of sources beyond {rfar:.1f} m skipped: together they add < 2^-24 f_0 to a receiver "
                    f"(DESIGN.md D8; CSF_FAR_EPS=0 evaluates every pair)")
        roof = {"bound": "valu", "kernel": kernel, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "launch_us": pair_s * 1e6, "launch_us_is": "median of the sampled launches", "launch_us_mean": mean_s["pair"] * 1e6,
                "launches_sampled": int(launches), "pairs_evaluated": evaluated,
                "pairs_all": pairs_all, "traffic": traffic, "traffic_source": traffic_src}
        if evaluated is not None and pair_s > 0:
            ops = OPS_PER_PAIR * evaluated + OPS_PER_TEST * work["tested"]
            roof["achieved"] = ops / pair_s / 1e12
            roof["frac"] = roof["achieved"] / VALU_PEAK_TFLOPS
            roof["field_only_frac"] = OPS_PER_PAIR * evaluated / pair_s / 1e12 / VALU_PEAK_TFLOPS   # crediting the field alone
            roof["work_per_launch"] = dict(work, op_equivalents=ops)
            roof["note"] = ("(100 fp32 op-equivalents x pairs evaluated + 25 x sources tested per lane) by one launch "
                            "(device counters) / median kernel duration; the stream of source records is served from "
                            "LDS/L2, HBM is not the roof (hbm_frac)")
        else:
            roof["achieved"] = roof["frac"] = None
            roof["note"] = "this engine's pair kernel does not count its evaluations"
        if traffic is not None and pair_s > 0:
            roof["hbm_frac"] = traffic / pair_s / 1e9 / HBM_PEAK_GBS
        if samples.size:
            roof["launch_us_min_med_max"] = [float(samples.min()), float(np.median(samples)), float(samples.max())]
        out = {
            "metric": "agent-steps/sec at N=16k TwoDBicycle", "value": value, "unit": "agent-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preroll_ticks": preroll,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{n} {args.model} agents, uniform random in {box:g} m x {box:g} m, "
                                   + ("every tracked pair evaluated" if not np.isfinite(rfar) else "all pairs within the 2^-24 far-field bound (every_pair: the literal sum)")
                                   + ", t_s=0.01" + (", curve-scenario road tiled on a 100 m grid" if road else ""),
                       "agents": n, "rider_model": args.model, "far_field": far_note,
                       "road_vertices": 0 if road is None else int(road[1].shape[0]),
                       "parallelism": f"index-sharded x{world}, RCCL all-gather of fp32 records per tick"
                       if world > 1 else "single GPU"},
            "healthy": healthy,
            "build_id": build_id(),
            "ranks": ranks,
            "timed_window": {"start": win0, "end": win1},
            "per_agent_launch": side_by_side,
            "dispersed": dispersed,
            "roofline": roof,
            "kernels_us": {"pair": pair_s * 1e6, "road": road_s * 1e6, "agent": agent_s * 1e6,
                           "all_gather": med_s["gather"] * 1e6, "tick": dt / args.steps * 1e6,
                           "are": "medians over the sampled launches (kernels_us_stats: median, min, max, mean, n); tick: wall time / steps",
                           "sampled_launches": {k: c for k, (_, c) in prof.items()}},
            "kernels_us_stats": stats,
            "algorithmic": {"bytes_per_launch": alg_bytes, "GBps": alg_bytes / pair_s / 1e9 if pair_s > 0 else None,
                            "x_hbm_peak": alg_bytes / pair_s / 1e9 / HBM_PEAK_GBS if pair_s > 0 else None,
                            "pairs_per_s": pairs_all / pair_s if pair_s > 0 else None,
                            "note": "SURVEY.md 8(d): 16 B x N sources per receiver over ALL pairs, evaluated or masked; "
                                    "a rate, not a fraction of a peak (the stream never leaves LDS/L2)"},
        }
        if world == 1 and not rehearse and np.isfinite(rfar) and args.every_pair_steps > 0:
            # the same population with the far-field cull off (reported beside the headline, never as `value`)
            # (timed like the headline: the read-backs above have let the clocks drop, so a scratch engine runs ~0.1 s first
            # and the population is then timed from tick 8)
            os.environ["CSF_FAR_EPS"] = "0"
            sx, ex = populate(), populate()
            ex.step(8, sync=True)
            sx.step(600, sync=True)
            t0 = time.perf_counter()
            ex.step(args.every_pair_steps, sync=True)
            dte = time.perf_counter() - t0
            ex.close()
            sx.close()
            del os.environ["CSF_FAR_EPS"]
            out["every_pair"] = {"value": n * args.every_pair_steps / dte, "unit": "agent-steps/s",
                                 "ms_per_step": dte / args.every_pair_steps * 1e3, "steps": args.every_pair_steps,
                                 "note": "CSF_FAR_EPS=0: no batch is skipped for distance"}
        if world == 1 and args.cpu_ticks > 0 and args.model == "twod" and road is None:
            out["cpu_baseline"] = cpu_baseline(n, box, args.cpu_ticks)
        if real_stdout is not None:
            sys.stdout.flush()
            os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
