#!/usr/bin/env python3
"""Where do the microseconds of the one-launch tick go?  (GPU box; measurement aid)

CSF_TRACE_AGENT makes every workgroup of mid_tick_kernel (csf_mid.hip) stamp wall_clock64() (100 MHz: 10 ns steps): wave 0 at
entry, when the destination-force phase is done, behind the barrier and at its end; wave 1 at entry, when its first receivers
have arrived and when its sums are done; wave 7 when its sums are done.  The engine writes the stamps of the LAST launch when
it is destroyed.  Each stamp waits for the loads / stores issued before it."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(model, n, box):
    import bench
    from cyclistsocialforce_amd import parameters
    from cyclistsocialforce_amd.engine import Engine

    s0, off, dq = bench.synthetic_population(n, box, reach=tuple(50.0 * k for k in range(1, 14)))
    if model == "invpend":
        s0 = np.c_[s0, np.zeros(n)]
    e = Engine(parameters.default_pod(model), n)
    e.add_agents(s0, 5.0)
    e.set_dest_queue(np.arange(n), off, dq, reset=True)
    e.step(300, sync=True)
    print("MID", e.mid_ticks())
    e.close()


def main():
    for model in ("twod", "invpend"):
        for n, box in ((64, 12.5), (1024, 200.0), (1024, 50.0), (2048, 70.7)):
            path = f"/tmp/mtrace_{model}_{n}.bin"
            env = dict(os.environ, CSF_TRACE_AGENT=path, CSF_MID_BELOW="4000")
            r = subprocess.run([sys.executable, __file__, "child", model, str(n), str(box)], env=env, capture_output=True, text=True)
            if "MID 300" not in r.stdout:
                print(r.stdout, r.stderr)
                raise SystemExit(1)
            w = np.fromfile(path, dtype=np.uint64).astype(np.int64)
            G = 4
            while G < 32 and (n + G - 1) // G > 256:
                G *= 2
            groups = (n + G - 1) // G
            w = w[:16 * groups].reshape(groups, 16)
            t0 = w[:, [0, 4]].min()
            us = lambda a: (a - t0) / 100.0
            print(f"{model} N={n} box={box}: {groups} workgroups of {G} road users; microseconds from the first wave's entry (mean / max over workgroups)")
            for k, name in ((0, "wave 0 entry"), (1, "wave 0 destination force done"), (4, "wave 1 entry"), (5, "wave 1 first sources loaded"), (6, "wave 1 sums done"),
                            (7, "wave 7 sums done"), (2, "wave 0 behind the barrier"), (3, "wave 0 end")):
                print(f"   {name:32s} {us(w[:, k]).mean():7.2f} {us(w[:, k]).max():7.2f}")
            print(f"   per workgroup: destination force {((w[:, 1] - w[:, 0]) / 100).mean():.2f}, pair sums (wave 1) {((w[:, 6] - w[:, 4]) / 100).mean():.2f}, "
                  f"rest of the tick {((w[:, 3] - w[:, 2]) / 100).mean():.2f}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]), float(sys.argv[4]))
    else:
        main()
