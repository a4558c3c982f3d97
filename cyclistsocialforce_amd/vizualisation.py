"""Matplotlib drawings for the vehicles of the engine — the visual hook of SURVEY.md §8(f)1.

The reference draws every vehicle from inside `vehicle.step()` (`vehicle.py:328, 695-732`, artists in
`vizualisation.py:25-430`; the module name is spelled as in the reference).  Here the numbers come back from the GPU
once per tick in one transfer, and `SocialForceIntersection._pull` calls `update_drawing()` of the vehicles that have
a drawing, so the artists are fed from the host mirror.  This module provides a compact drawing of its own — body
outline, past trajectory, destinations, resulting-force arrow — with the interface the reference's hook expects:
`update(vehicle, Fdest=, Frep=, Fres=)` and `set_animated(flag)`.  Matplotlib is imported when a drawing is created.
"""
import numpy as np


class VehicleDrawing:
    """A road user on a matplotlib axes: oriented body, trajectory so far, remaining destinations, force arrow."""

    LENGTH, WIDTH = 1.8, 0.6              # metres, a bicycle seen from above

    def __init__(self, ax, vehicle, animated=False, color=None, draw_trajectory=True, draw_destinations=True,
                 draw_force=True, force_scale=0.3):
        from matplotlib.lines import Line2D
        from matplotlib.patches import Polygon

        self.ax = ax
        self.force_scale = force_scale
        color = color or ax._get_lines.get_next_color()
        self.body = Polygon(self._outline(vehicle._s), closed=True, facecolor=color, edgecolor="black", linewidth=0.5,
                            animated=animated, zorder=3)
        ax.add_patch(self.body)
        self.trajectory = self.destinations = self.force = None
        if draw_trajectory:
            self.trajectory = Line2D([], [], color=color, linewidth=0.8, animated=animated, zorder=2)
            ax.add_line(self.trajectory)
        if draw_destinations:
            self.destinations = Line2D([], [], color=color, marker="x", linestyle="none", markersize=4,
                                       animated=animated, zorder=1)
            ax.add_line(self.destinations)
        if draw_force:
            self.force = Line2D([], [], color="black", linewidth=0.8, animated=animated, zorder=4)
            ax.add_line(self.force)
        self.update(vehicle)

    def _outline(self, s):
        c, n = np.cos(s[2]), np.sin(s[2])
        front, rear, half = 0.6 * self.LENGTH, -0.4 * self.LENGTH, 0.5 * self.WIDTH
        local = np.array([[front, 0.0], [0.2 * front, half], [rear, half], [rear, -half], [0.2 * front, -half]])
        rot = np.array([[c, -n], [n, c]])
        return local @ rot.T + np.asarray(s[:2], dtype=float)

    def artists(self):
        return [a for a in (self.body, self.trajectory, self.destinations, self.force) if a is not None]

    def update(self, vehicle, Fdest=None, Frep=None, Fres=None):
        """the hook of vehicle.py:722-732"""
        s = vehicle._s
        self.body.set_xy(self._outline(s))
        if self.trajectory is not None:
            i = int(vehicle.i)
            self.trajectory.set_data(vehicle.traj[0, : i + 1], vehicle.traj[1, : i + 1])
        if self.destinations is not None and vehicle.destqueue is not None:
            rest = vehicle.destqueue[int(vehicle.destpointer):]
            self.destinations.set_data(rest[:, 0], rest[:, 1])
        if self.force is not None:
            F = Fres if Fres is not None else getattr(vehicle, "force", (0.0, 0.0))
            self.force.set_data([s[0], s[0] + self.force_scale * F[0]], [s[1], s[1] + self.force_scale * F[1]])

    def set_animated(self, animated):
        for a in self.artists():
            a.set_animated(animated)

    def remove(self):
        for a in self.artists():
            a.remove()


BicycleDrawing2D = VehicleDrawing       # vizualisation.py:564: the name scenario scripts import


def plot_states(vehicle, t_end=None, axes=None):
    """vehicle.py:734-787: the stored trajectory of every state over time, one row of axes per state."""
    import matplotlib.pyplot as plt

    n_states = vehicle.traj.shape[0]
    if axes is None:
        _, axes = plt.subplots(n_states, 1, sharex=True)
    axes = np.atleast_1d(axes)
    i_end = int(vehicle.i) if t_end is None else min(int(vehicle.i), int(t_end / vehicle.params.t_s))
    t = np.arange(i_end + 1) * vehicle.params.t_s
    for k in range(n_states):
        axes[k].plot(t, vehicle.traj[k, : i_end + 1], label=str(vehicle.id))
        axes[k].set_ylabel(vehicle.s_names[k] if k < len(vehicle.s_names) else "")
    axes[-1].set_xlabel("t [s]")
    return axes


def plot_forces(vehicle, t_end=None, axes=None, components_to_plot=("magnitude", "direction")):
    """vehicle.py:789-917: the stored resulting force (needs saveForces=True) as magnitude / direction / x / y."""
    import matplotlib.pyplot as plt

    if not vehicle.saveForces:
        raise ValueError("plot_forces needs a vehicle created with saveForces=True")
    comps = list(components_to_plot)
    if axes is None:
        _, axes = plt.subplots(len(comps), 1, sharex=True)
    axes = np.atleast_1d(axes)
    i_end = int(vehicle.i) if t_end is None else min(int(vehicle.i), int(t_end / vehicle.params.t_s))
    t = np.arange(1, i_end + 1) * vehicle.params.t_s
    Fx, Fy = vehicle.trajF[0, 1 : i_end + 1], vehicle.trajF[1, 1 : i_end + 1]
    series = {"magnitude": np.hypot(Fx, Fy), "direction": np.arctan2(Fy, Fx), "x": Fx, "y": Fy}
    for ax, name in zip(axes, comps):
        if name not in series:
            raise ValueError(f"components_to_plot must be drawn from {list(series)}")
        ax.plot(t, series[name], label=str(vehicle.id))
        ax.set_ylabel(f"F {name}")
    axes[-1].set_xlabel("t [s]")
    return axes
