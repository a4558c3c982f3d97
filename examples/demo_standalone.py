#!/usr/bin/env python3
"""The reference's stand-alone demo (demo/demoCSFstandalone.py:94-146: three cyclists in an encroachment conflict in
open space, 7 s of simulated time) on the MI355X engine, with the reference's own class names and call sequence.
Only the import lines differ; animation is replaced by a printed summary."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cyclistsocialforce_amd.intersection import SocialForceIntersection  # noqa: E402
from cyclistsocialforce_amd.scenario import Scenario  # noqa: E402
from cyclistsocialforce_amd.vehicle import (Bicycle, InvPendulumBicycle, PlanarPointBicycle,  # noqa: E402
                                            TwoDBicycle)

MODELS = {"2d": TwoDBicycle, "planartwowheel": Bicycle, "invpendulum": InvPendulumBicycle,
          "planarpoint": PlanarPointBicycle}


class Demo(Scenario):
    def __init__(self, cls):
        a = cls((-23 + 17, 0, 0, 5, 0, 0, 0, 0), id="a", saveForces=True)
        a.params.v_desired_default = 4.5
        b = cls((0 + 15, -20, np.pi / 2, 5, 0, 0, 0, 0), id="b", saveForces=True)
        b.params.v_desired_default = 5.0
        c = cls((-2 + 15, -20, np.pi / 2, 5, 0, 0, 0, 0), id="c", saveForces=True)
        c.params.v_desired_default = 5.0
        a.setDestinations((35, 64, 65), (0, 0, 0))
        b.setDestinations((15, 15, 15), (20, 49, 50))
        c.setDestinations((13, 13, 13), (20, 49, 50))
        self.bikes = (a, b, c)
        self.intersection = SocialForceIntersection(self.bikes)
        Scenario.__init__(self, self.step_func, t_r=0, verbose=False)

    def step_func(self):
        self.intersection.step()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="2d", choices=sorted(MODELS))
    ap.add_argument("--t-end", type=float, default=7.0)
    args = ap.parse_args()
    demo = Demo(MODELS[args.model])
    t0 = time.perf_counter()
    demo.run(args.t_end)
    dt = time.perf_counter() - t0
    ticks = len(demo.intersection.hist_n_vecs)
    print(f"{args.model}: {ticks} ticks in {dt * 1e3:.1f} ms including engine start-up ({dt / ticks * 1e6:.0f} us per tick)")
    for v in demo.bikes:
        print(f"  {v.id}: x = {v.s[0]:9.4f}  y = {v.s[1]:9.4f}  psi = {v.s[2]:8.4f}  v = {v.s[3]:7.4f}   "
              f"|F| of the last tick = {v.F[-1]:.4f}   (traj columns 0..{v.i} hold the history)")
