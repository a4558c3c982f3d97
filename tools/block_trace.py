#!/usr/bin/env python3
"""Workgroup timeline of the pair kernel (measurement aid).

    CSF_TRACE_BLOCKS=/tmp/trace.bin python3 tools/block_trace.py run      # GPU box: 60 ticks of the bench population
    python3 tools/block_trace.py show gpurun_out/trace.bin                # anywhere: summarise

Every wave of pair_cull_kernel stores wall_clock64() (100 MHz) at entry and exit plus HW_ID / XCC_ID; the engine
writes the records of the LAST launch to the file when it is destroyed.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import bench
    from cyclistsocialforce_amd import parameters
    from cyclistsocialforce_amd.engine import Engine

    n = int(os.environ.get("AGENTS", "16384"))
    s0, off, dq = bench.synthetic_population(n, 200.0)
    eng = Engine(parameters.default_pod("twod"), n)
    eng.add_agents(s0, 5.0)
    eng.set_dest_queue(np.arange(n), off, dq, reset=True)
    eng.step(int(os.environ.get("TICKS", "60")), sync=True)
    eng.close()


def show(path):
    w = np.fromfile(path, dtype=np.uint64).reshape(-1, 3)
    if len(sys.argv) > 3:   # source chunks of the launch (16 at N = 16 384): when does each chunk's work run, how long is it
        nch = int(sys.argv[3])
        used = np.nonzero(w[:, 1] > 0)[0]
        per = (used.max() + 1 + nch - 1) // nch
        b = int(w[used, 0].min())
        print("chunk  waves  first_entry_us  last_exit_us  wave_duration_mean_us  wave-microseconds")
        for c in range(nch):
            k = used[(used >= c * per) & (used < (c + 1) * per)]
            a0, a1 = (w[k, 0].astype(np.int64) - b) / 100.0, (w[k, 1].astype(np.int64) - b) / 100.0
            print(f"{c:5d}  {len(k):5d}  {a0.min():8.1f}  {a1.max():8.1f}  {(a1 - a0).mean():8.2f}  {(a1 - a0).sum():10.0f}")
    w = w[w[:, 1] > 0]
    t0, t1, hw = w[:, 0].astype(np.int64), w[:, 1].astype(np.int64), w[:, 2]
    base = t0.min()
    t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0      # microseconds
    dur = t1 - t0
    end = t1.max()
    print(f"waves {len(w)}  kernel span {end:.1f} us  wave duration mean {dur.mean():.2f} p50 {np.median(dur):.2f} "
          f"p95 {np.percentile(dur, 95):.2f} max {dur.max():.2f} us")
    xcc = (hw >> np.uint64(32)) & np.uint64(0xF)
    hwid = hw & np.uint64(0xFFFFFFFF)
    cu = (hwid >> np.uint64(8)) & np.uint64(0xF)
    sh = (hwid >> np.uint64(12)) & np.uint64(0x1)
    se = (hwid >> np.uint64(13)) & np.uint64(0x7)
    simd = (hwid >> np.uint64(4)) & np.uint64(0x3)
    unit = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print(f"distinct xcc {len(np.unique(xcc))}  CUs {len(np.unique(unit))}  SIMDs {len(np.unique(unit * 4 + simd))}")
    # resident waves over time
    edges = np.linspace(0, end, 41)
    print("time_us  resident_waves  (of %d slots at 6 waves/SIMD)" % (len(np.unique(unit * 4 + simd)) * 6))
    for a, b in zip(edges[:-1], edges[1:]):
        mid = 0.5 * (a + b)
        print(f"{mid:7.1f}  {int(np.sum((t0 <= mid) & (t1 > mid))):6d}")
    # busy time per SIMD: how unevenly does the work end
    last = np.zeros(int((unit * 4 + simd).max()) + 1)
    np.maximum.at(last, (unit * 4 + simd).astype(np.int64), t1)
    last = last[last > 0]
    print(f"per-SIMD last exit: min {last.min():.1f} p10 {np.percentile(last, 10):.1f} p50 {np.median(last):.1f} "
          f"p90 {np.percentile(last, 90):.1f} max {last.max():.1f} us; mean idle at the end {np.mean(end - last):.1f} us")
    first = np.full(int((unit * 4 + simd).max()) + 1, np.inf)
    np.minimum.at(first, (unit * 4 + simd).astype(np.int64), t0)
    first = first[np.isfinite(first)]
    print(f"per-SIMD first entry: p50 {np.median(first):.1f} p90 {np.percentile(first, 90):.1f} max {first.max():.1f} us")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        show(sys.argv[2])
