#!/usr/bin/env python3
"""Per-rank evidence for the configs BASELINE defines on eight GPUs (4: 262 144 TwoDBicycle, 5: 1 048 576 PlanarPointBicycle + road),
gathered on ONE device (no 8-GPU node is ours to launch on).  Two modes, one JSON line per result:

  tools/shard_ranks.py CONFIG [--world 8] [--ticks K]
      every rank r of `world` in turn (CSF_FAKE_SHARD=r/world: only that rank's receiver block is computed; one process, the
      population built once): tick (wall), pair / road / per-agent kernel (median, min, max over sampled launches), the pairs its
      block evaluates - the IMBALANCE between ranks is the number; then the unsharded engine the same way; then the summary:
      slowest rank, the emulated compute-only ceiling, and the exchange a real run adds (32 B per road user and tick over 7 xGMI
      links of ~153 GB/s each).
  tools/shard_ranks.py CONFIG --loopback 2 [--ticks 130]
      a 2-way loopback group (csf_comm_init_loopback: the code path of csf_comm_init with device copies for the collective), to
      be run under `rocprofv3 --kernel-trace --stats`: its kernel_stats.csv holds what a rank REPEATS whatever the world size -
      the copy of all n exchange records into binned order every tick (sorted_copy_kernel), and the re-binning of all n records
      (keys, radix sort, rebase, circles, candidate lists) every 64 ticks.  tools/shard_summary.py puts the two together."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from large_configs import CONFIGS, build  # noqa: E402
from cyclistsocialforce_amd.engine import Engine  # noqa: E402

XGMI_LINK_GBS, XGMI_LINKS = 153.0, 7


def timed(e, ticks, warm):
    e.step(warm, sync=True)
    e.profile(max(1, ticks // 32))
    t0 = time.perf_counter()
    e.step(ticks, sync=True)
    dt = time.perf_counter() - t0
    stats = e.profile_stats()
    e.profile_kernels()
    e.profile(0)
    return dt / ticks * 1e6, stats


def main():
    args = sys.argv[1:]
    key = args[0]
    cfg = dict(CONFIGS[key])
    world = int(args[args.index("--world") + 1]) if "--world" in args else 8
    ticks = int(args[args.index("--ticks") + 1]) if "--ticks" in args else {"4": 24, "5": 8}.get(key, 200)
    n = cfg["n"]
    if "--loopback" in args:
        w = int(args[args.index("--loopback") + 1])
        members = [build(cfg)[0] for _ in range(w)]
        Engine.loopback_group(members)
        Engine.step_group(members, ticks, sync=True)
        print(json.dumps({"config": cfg["name"], "loopback_world": w, "ticks": ticks, "agents": n}), flush=True)
        for m in members:
            m.close()
        return
    rows = []
    for r in list(range(world)) + [None]:
        if r is None:
            os.environ.pop("CSF_FAKE_SHARD", None)
        else:
            os.environ["CSF_FAKE_SHARD"] = f"{r}/{world}"
        e, road = build(cfg)                                  # (the knobs are read at csf_create)
        tick_us, stats = timed(e, ticks, cfg["warm"] + 1)
        ev, _ = e.count_pairs()
        med = lambda k: None if stats[k] is None else stats[k]["median"]   # noqa: E731
        row = {"config": cfg["name"], "rank": "unsharded" if r is None else f"{r}/{world}", "tick_us": tick_us, "pair_us": med("pair"),
               "road_us": med("road"), "agent_us": med("agent"), "kernels_us_stats": stats, "pairs_evaluated": ev, "ticks": ticks}
        print(json.dumps(row), flush=True)
        rows.append(row)
        e.close()
    ranks, whole = rows[:-1], rows[-1]
    slow = max(ranks, key=lambda x: x["tick_us"])
    xbytes = 32.0 * n                                         # the exchange records of all ranks: what every rank receives per tick
    wire_us = xbytes * (world - 1) / world / (XGMI_LINKS * XGMI_LINK_GBS * 1e9) * 1e6
    print(json.dumps({"config": cfg["name"], "summary": True, "world": world, "unsharded_tick_us": whole["tick_us"],
                      "rank_tick_us": [x["tick_us"] for x in ranks], "slowest_rank": slow["rank"], "slowest_rank_tick_us": slow["tick_us"],
                      "imbalance_slowest_over_mean": slow["tick_us"] / float(np.mean([x["tick_us"] for x in ranks])),
                      "compute_only_ceiling_x": whole["tick_us"] / slow["tick_us"],
                      "pairs_evaluated_by_rank": [x["pairs_evaluated"] for x in ranks],
                      "exchange_bytes_per_tick": xbytes, "exchange_wire_us_at_7_links": wire_us,
                      "note": "CSF_FAKE_SHARD ranks run pair + road + per-agent kernels of their block and the unsharded re-binning; what a real "
                              "rank adds - the binned copy of all n gathered records per tick, the all-gather itself - is in the loopback "
                              "trace (tools/shard_summary.py)"}), flush=True)


if __name__ == "__main__":
    main()
