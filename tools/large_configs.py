#!/usr/bin/env python3
"""BASELINE.json configs 2-5 on ONE GPU (configs 4/5 are specified for 8 GPUs; here they check that the
single-device path holds at those sizes and give per-tick times).  Prints one JSON line per config."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synthetic_population, tiled_curve_road  # noqa: E402
from cyclistsocialforce_amd import parameters  # noqa: E402
from cyclistsocialforce_amd.engine import Engine  # noqa: E402

LONG_REACH = tuple(50.0 * k for k in range(1, 14))   # waypoints every 50 m out to 650 m: outlasts 10 000 ticks at 5 m/s


def run(name, model, n, box, ticks, road=None, warm=2, reach=(50.0, 99.0, 100.0)):
    s0, off, dq = synthetic_population(n, box, reach=reach)
    if model == "invpend":
        s0 = np.c_[s0, np.zeros(n)]
    elif model == "planarpoint":
        s0 = s0[:, :4]
    e = Engine(parameters.default_pod(model), n)
    e.add_agents(s0, 5.0)
    e.set_dest_queue(np.arange(n), off, dq, reset=True)
    if road is not None:
        e.set_road(*road)
    e.step(warm, sync=True)
    if n < 2048:
        # (populations that tick in one launch - csf_mid.hip - take the two launches when kernels are timed one by one: the
        # rate first, without time stamps; then a short sampled stretch for the two kernels' own times)
        t0 = time.perf_counter()
        e.step(ticks, sync=True)
        dt = time.perf_counter() - t0
        e.profile(8)
        e.step(512, sync=True)
    else:
        e.profile(max(1, ticks // 64))
        t0 = time.perf_counter()
        e.step(ticks, sync=True)
        dt = time.perf_counter() - t0
    prof = {k: ms * 1e3 / max(c, 1) for k, (ms, c) in e.profile_kernels().items()}     # mean microseconds per launch
    e.profile(0)
    s = e.state()
    st = e.status()
    evaluated, kernel = e.count_pairs()
    print(json.dumps({"config": name, "model": model, "agents": n, "box_m": box, "ticks": ticks,
                      "ms_per_tick": dt / ticks * 1e3, "agent_steps_per_s": n * ticks / dt,
                      "finite": bool(np.isfinite(s).all()), "status_flags": int((st != 0).sum()),
                      "road_vertices": 0 if road is None else int(road[1].shape[0]),
                      "pair_kernel": kernel, "pair_us": prof["pair"], "road_us": prof["road"],
                      "agent_us": prof["agent"], "pairs_evaluated": evaluated, "one_launch_ticks": e.mid_ticks()}), flush=True)
    e.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["2", "3", "4", "5"]
    if "2" in which:
        # (warm-up past the first re-binning at tick 64: the first launch of every kernel loads its code - milliseconds, once per process)
        run("2: 1,024 TwoDBicycle, 10,000 steps", "twod", 1024, 200.0, 10000, reach=LONG_REACH, warm=int(os.environ.get("CSF_WARM_TICKS", "70")))
    if "3" in which:
        run("3: 16,384 InvertedPendulumBicycle", "invpend", 16384, 200.0, 1000)
    if "4" in which:
        run("4: 262,144 TwoDBicycle (single GPU)", "twod", 262144, 800.0, 20, warm=1)
    if "5" in which:
        run("5: 1,048,576 PlanarPointBicycle + road (single GPU)", "planarpoint", 1048576, 1600.0, 10,
            road=tiled_curve_road(1600.0), warm=1)
