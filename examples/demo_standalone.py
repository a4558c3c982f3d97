#!/usr/bin/env python3
"""The reference's stand-alone demo (demo/demoCSFstandalone.py:94-146: three cyclists in an encroachment conflict in
open space, 7 s of simulated time) on the MI355X engine, with the reference's own class names and call sequence.
Only the import lines differ.  `--animate` draws the scene as the reference does (matplotlib, blitting; with `--save FILE`
on the Agg canvas, writing the final frame and the state / force histories instead of opening windows)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cyclistsocialforce_amd.intersection import SocialForceIntersection  # noqa: E402
from cyclistsocialforce_amd.scenario import Scenario  # noqa: E402
from cyclistsocialforce_amd.vehicle import (BalancingRiderBicycle, Bicycle, InvPendulumBicycle, PlanarBicycle,  # noqa: E402
                                            PlanarPointBicycle, TwoDBicycle)

MODELS = {"balancingrider": BalancingRiderBicycle, "2d": TwoDBicycle, "planartwowheel": Bicycle, "invpendulum": InvPendulumBicycle,
          "planarpoint": PlanarPointBicycle, "planarbike": PlanarBicycle}      # (balancingrider: the reference demo's default)


class Demo(Scenario):
    def __init__(self, cls, ax=None):
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
        self.intersection = SocialForceIntersection(self.bikes, animate=ax is not None, axes=ax)
        Scenario.__init__(self, self.step_func, t_r=0, verbose=False, animate=ax is not None, axes=ax)

    def step_func(self):
        self.intersection.step()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="2d", choices=sorted(MODELS))
    ap.add_argument("--t-end", type=float, default=7.0)
    ap.add_argument("--animate", action="store_true", help="draw the scene (demoCSFstandalone.py:120-135)")
    ap.add_argument("--save", default=None, help="with --animate: file stem for the final frame and the history plots")
    args = ap.parse_args()
    ax = None
    if args.animate:
        import matplotlib
        if args.save:
            matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        fig, ax = plt.subplots(1, 1)
        ax.set_title(f"Interaction demo: {args.model}")
        ax.set_xlim(0, 30)
        ax.set_ylim(-10, 20)
        ax.set_aspect("equal")
    demo = Demo(MODELS[args.model], ax)
    t0 = time.perf_counter()
    demo.run(args.t_end)
    dt = time.perf_counter() - t0
    ticks = len(demo.intersection.hist_n_vecs)
    print(f"{args.model}: {ticks} ticks in {dt * 1e3:.1f} ms including engine start-up ({dt / ticks * 1e6:.0f} us per tick)")
    for v in demo.bikes:
        print(f"  {v.id}: x = {v.s[0]:9.4f}  y = {v.s[1]:9.4f}  psi = {v.s[2]:8.4f}  v = {v.s[3]:7.4f}   "
              f"|F| of the last tick = {v.F[-1]:.4f}   (traj columns 0..{v.i} hold the history)")
    if args.animate:
        demo.intersection.set_animated(False)                   # demoCSFstandalone.py:148-149
        axes_states = axes_forces = None
        for bike in demo.intersection.vehicles:                 # demoCSFstandalone.py:152-156
            axes_states = bike.plot_states(t_end=args.t_end, axes=axes_states)
            axes_forces = bike.plot_forces(t_end=args.t_end, axes=axes_forces, components_to_plot=["magnitude", "direction"])
        if args.save:
            fig.savefig(args.save + "_scene.png")
            axes_states[0].get_figure().savefig(args.save + "_states.png")
            axes_forces[0].get_figure().savefig(args.save + "_forces.png")
        else:
            plt.show(block=True)
