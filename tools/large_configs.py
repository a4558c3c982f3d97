#!/usr/bin/env python3
"""BASELINE.json configs 1-5 on ONE GPU (configs 4/5 are specified for 8 GPUs; here they check that the single-device path
holds at those sizes and give per-tick times).  One JSON line per config, each with the `roofline` object of bench.py for the
config's dominant kernel: (100 op-equivalents x pairs evaluated + 25 x sources tested) per launch - device counters where the
kernel counts (csf_count_pairs), the reference's own mask counted on the host where it does not (the one-launch tick and the
one-wave kernel put every source through the test) - over the kernel's MEDIAN duration, against the fp32 vector peak.

    tools/large_configs.py [1 2 3 4 5] [--ticks K] [--plain]
--plain: only build the config and step it (the program rocprofv3 runs: tools/profile_configs.sh), then print the work counts."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import OPS_PER_PAIR, OPS_PER_TEST, VALU_PEAK_TFLOPS, synthetic_population, tiled_curve_road  # noqa: E402
from cyclistsocialforce_amd import parameters  # noqa: E402
from cyclistsocialforce_amd.engine import Engine  # noqa: E402

LONG_REACH = tuple(50.0 * k for k in range(1, 14))   # waypoints every 50 m out to 650 m: outlasts 10 000 ticks at 5 m/s

CONFIGS = {
    "1": dict(name="1: demoCSFstandalone geometry, 3 TwoDBicycle", model="twod", n=3, box=70.0, ticks=700, warm=10, demo=True),
    # (warm-up past the first re-binning at tick 64: the first launch of every kernel loads its code - milliseconds, once per process)
    "2": dict(name="2: 1,024 TwoDBicycle, 10,000 steps", model="twod", n=1024, box=200.0, ticks=10000, warm=70, reach=LONG_REACH),
    "3": dict(name="3: 16,384 InvertedPendulumBicycle", model="invpend", n=16384, box=200.0, ticks=1000, warm=2),
    "4": dict(name="4: 262,144 TwoDBicycle (single GPU)", model="twod", n=262144, box=800.0, ticks=20, warm=1),
    "5": dict(name="5: 1,048,576 PlanarPointBicycle + road (single GPU)", model="planarpoint", n=1048576, box=1600.0, ticks=10, warm=1, road=True),
}


def build(cfg):
    """the engine of a BASELINE config, populated; config 1: demo/demoCSFstandalone.py:101-118 from the golden file"""
    model, n = cfg["model"], cfg["n"]
    if cfg.get("demo"):
        g = np.load(os.path.join(ROOT, "tests", "golden", "trajectories.npz"))
        s0, vdes, off, dq = g["demo_twod_s0"], g["demo_twod_vdes"], g["demo_twod_off"], g["demo_twod_dq"]
    else:
        s0, off, dq = synthetic_population(n, cfg["box"], reach=cfg.get("reach", (50.0, 99.0, 100.0)))
        vdes = 5.0
        if model == "invpend":
            s0 = np.c_[s0, np.zeros(n)]
        elif model == "planarpoint":
            s0 = s0[:, :4]
    road = tiled_curve_road(cfg["box"]) if cfg.get("road") else None
    e = Engine(parameters.default_pod(model), n)
    e.add_agents(s0, vdes)
    e.set_dest_queue(np.arange(n), off, dq, reset=True)
    if road is not None:
        e.set_road(*road)
    return e, road


def work_of(e, n):
    """(pairs evaluated, sources tested, kernel name, how counted) of ONE pair evaluation on the current snapshot"""
    work, kernel = e.count_pairs(detail=True)
    if work is not None:
        return int(work["evaluated"]), int(work["tested"]), kernel, "device counters (csf_count_pairs)"
    # the kernels that do not count put every source of every receiver through the test; what they evaluate is what the
    # reference's mask tracks (intersection.py:690-745): counted on the host from csf_untracked
    if n <= 4096:
        U = e.untracked()
        return int((U == 0).sum()), n * (n - 1), kernel, "host: csf_untracked (every source tested, tracked pairs evaluated)"
    return None, None, kernel, "not counted"


def roofline(kernel, evaluated, tested, stats, how):
    roof = {"bound": "valu", "kernel": kernel, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "achieved": None, "frac": None,
            "pairs_evaluated": evaluated, "sources_tested": tested, "counted": how, "traffic": None}
    if stats is not None:
        roof["launch_us"] = stats["median"]
        roof["launch_us_min_med_max"] = [stats["min"], stats["median"], stats["max"]]
        roof["launches_sampled"] = stats["n"]
        if evaluated is not None and stats["median"] > 0:
            ops = OPS_PER_PAIR * evaluated + OPS_PER_TEST * tested
            roof["achieved"] = ops / (stats["median"] * 1e-6) / 1e12
            roof["frac"] = roof["achieved"] / VALU_PEAK_TFLOPS
            roof["field_only_frac"] = OPS_PER_PAIR * evaluated / (stats["median"] * 1e-6) / 1e12 / VALU_PEAK_TFLOPS
    return roof


def run(key, ticks=None, plain=False):
    cfg = CONFIGS[key]
    n = cfg["n"]
    ticks = ticks or cfg["ticks"]
    e, road = build(cfg)
    e.step(cfg["warm"], sync=True)
    out = {"config": cfg["name"], "model": cfg["model"], "agents": n, "box_m": cfg["box"], "ticks": ticks,
           "road_vertices": 0 if road is None else int(road[1].shape[0])}
    if plain:
        # (what rocprofv3 profiles: the engine's own choice of kernels, nothing sampled by the engine itself; calls of 100 ticks so
        # that the one-wave kernel - one launch per CALL - shows up as more than one launch)
        t0 = time.perf_counter()
        for _ in range(max(1, ticks // 100)):
            e.step(min(ticks, 100))
        e.sync()
        dt = time.perf_counter() - t0
        done = max(1, ticks // 100) * min(ticks, 100)
        ev, te, kernel, how = work_of(e, n)
        out.update({"ms_per_tick": dt / done * 1e3, "ticks": done, "pair_kernel": kernel, "pairs_evaluated": ev, "sources_tested": te, "counted": how,
                    "one_launch_ticks": e.mid_ticks(), "one_wave_ticks": e.small_ticks()})
        print(json.dumps(out), flush=True)
        e.close()
        return
    fused = n < 2048
    if fused:
        # (populations that tick in one launch - csf_mid.hip, the one-wave kernel - take the two launches when kernels are timed one by
        # one: the rate first, without time stamps; then a short sampled stretch for the two kernels' own times)
        t0 = time.perf_counter()
        e.step(ticks, sync=True)
        dt = time.perf_counter() - t0
        mid, small = e.mid_ticks(), e.small_ticks()
        e.profile(8)
        e.step(512, sync=True)
    else:
        e.profile(max(1, ticks // 64))
        t0 = time.perf_counter()
        e.step(ticks, sync=True)
        dt = time.perf_counter() - t0
        mid, small = e.mid_ticks(), e.small_ticks()
    stats = e.profile_stats()
    e.profile_kernels()
    e.profile(0)
    s = e.state()
    st = e.status()
    ev, te, kernel, how = work_of(e, n)
    med = lambda k: None if stats[k] is None else stats[k]["median"]   # noqa: E731
    roof = roofline(kernel, ev, te, stats["pair"], how)
    if fused:
        roof["note"] = ("this population ticks in ONE launch per tick (or per call); the pair kernel timed here is the one the engine takes when "
                        "kernels are sampled one by one - the fused launch's own duration is in profiles/*_kernel_stats.csv (rocprofv3)")
    out.update({"ms_per_tick": dt / ticks * 1e3, "agent_steps_per_s": n * ticks / dt, "finite": bool(np.isfinite(s).all()),
                "status_flags": int((st != 0).sum()), "pair_kernel": kernel, "pair_us": med("pair"), "road_us": med("road"),
                "agent_us": med("agent"), "kernels_us_stats": stats, "pairs_evaluated": ev, "one_launch_ticks": mid, "one_wave_ticks": small,
                "roofline": roof})
    print(json.dumps(out), flush=True)
    e.close()


if __name__ == "__main__":
    args = sys.argv[1:]
    plain = "--plain" in args
    ticks = None
    if "--ticks" in args:
        k = args.index("--ticks")
        ticks = int(args[k + 1])
        del args[k:k + 2]                               # (its value is not a config number)
    which = [a for a in args if a in CONFIGS] or ["2", "3", "4", "5"]
    for key in which:
        run(key, ticks, plain)
