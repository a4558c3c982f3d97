#!/usr/bin/env python3
"""Where do the microseconds of the per-agent kernel go?  (GPU box; measurement aid)

    python3 tools/agent_timeline.py            # Bicycle, TwoDBicycle, InvPendulumBicycle at N = 3, 1 024, 16 384
    python3 tools/agent_timeline.py balancingrider planarpoint 16384      # other classes / sizes

CSF_TRACE_AGENT makes every wave of agent_kernel stamp wall_clock64() (100 MHz: 10 ns steps) at entry, when its own
scalars, partial sums and road term have arrived, after the destination force, after the combine phase, after the
controller + kinematics, when its stores are issued and when they are done; the engine writes the stamps of the LAST
launch when it is destroyed.  Each stamp waits for the loads / stores issued before it, so the traced kernel is a little
slower than the product's (its untraced duration, from the dispatch's own time stamps, is printed beside it)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(model, n, path):
    import bench
    from cyclistsocialforce_amd import parameters
    from cyclistsocialforce_amd.engine import Engine

    box = max(10.0, (n / 0.41) ** 0.5)
    s0, off, dq = bench.synthetic_population(n, box)
    from cyclistsocialforce_amd import _ffi, engine
    width = _ffi.N_STATES[engine.MODEL_IDS[model]]
    s0 = np.c_[s0, np.zeros((n, max(0, width - s0.shape[1])))][:, :width]
    e = Engine(parameters.default_pod(model), n)
    e.add_agents(s0, 5.0)
    e.set_dest_queue(np.arange(n), off, dq, reset=True)
    e.step(200, sync=True)
    e.profile(1)
    e.profile_kernels()
    e.step(64, sync=True)
    k = e.profile_kernels()
    print(f"KERNEL_US {k['agent'][0] / max(k['agent'][1], 1) * 1e3:.2f} {k['pair'][0] / max(k['pair'][1], 1) * 1e3:.2f}")
    e.close()


def main():
    names = ["own scalars + partial sums arrive", "destination force", "combine", "controller + kinematics", "stores issued", "stores done"]
    print("per-agent kernel: mean over the waves of the last launch, microseconds (wall_clock64, 10 ns steps)")
    models = [a for a in sys.argv[1:] if not a.isdigit()] or ["bicycle", "twod", "invpend"]
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [3, 1024, 16384]
    for model in models:
        for n in sizes:
            path = f"/tmp/atrace_{model}_{n}.bin"
            out = {}
            for traced in (False, True):
                env = dict(os.environ)
                env.pop("CSF_TRACE_AGENT", None)
                if traced:
                    env["CSF_TRACE_AGENT"] = path
                r = subprocess.run([sys.executable, __file__, "child", model, str(n), path], env=env, capture_output=True, text=True)
                line = [l for l in r.stdout.splitlines() if l.startswith("KERNEL_US")]
                if not line:
                    print(r.stdout, r.stderr)
                    raise SystemExit(1)
                out[traced] = [float(v) for v in line[0].split()[1:]]
            w = np.fromfile(path, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
            w = w[w[:, 7] > 0]
            t = (w - w[:, :1]) / 100.0
            seg = np.c_[t[:, 1], t[:, 2] - t[:, 1], t[:, 4] - t[:, 2], t[:, 5] - t[:, 4], t[:, 6] - t[:, 5], t[:, 7] - t[:, 6]]
            span = (w[:, 7].max() - w[:, 0].min()) / 100.0
            late = (w[:, 0].max() - w[:, 0].min()) / 100.0
            print(f"{model:8s} N={n:6d}: kernel {out[False][0]:6.2f} us untraced, {out[True][0]:6.2f} traced (pair kernel {out[False][1]:.1f}); "
                  f"{len(w)} waves, first entry -> last exit {span:.2f}, last wave enters {late:.2f} after the first")
            print("            " + "; ".join(f"{nm} {seg[:, i].mean():.2f}" for i, nm in enumerate(names)) + f"; a wave's life {t[:, 7].mean():.2f} (max {t[:, 7].max():.2f})")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]), sys.argv[4])
    else:
        main()
