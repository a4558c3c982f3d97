"""Tick driver with the interface of `cyclistsocialforce.scenario.Scenario` (scenario.py:53-265): same constructor
arguments, `run(t_end)`, `reset()`, the public counters `i`, `t`, `i_end`, and the same three behaviours a caller can
observe —

* `run(t_end)` executes `step_func()` until `i == int(t_end / t_s)`; it can be called again with a later `t_end`;
* every tick is stretched to at least `t_r` seconds of wall time (real-time pacing; `t_r=0` runs flat out);
* with `verbose=True` the run waits for <Enter> first and keeps one status line up to date.

Written from that description, not from the reference module, which cannot be imported without sumolib, traci, cv2
and mypyutils (scenario.py:13-50).  Additions: `hist_run_time` (seconds each tick took before pacing) and
`ticks_per_second()`.  Animation (`animate=True`) draws through the vehicles' drawing hooks with matplotlib when it is
installed; writing videos is out of scope.
"""
import sys
import time
from datetime import timedelta


class _RealTimePacer:
    """Stretches ticks to a minimum wall-clock duration."""

    def __init__(self, period):
        self.period = float(period)

    def remaining(self, started):
        return max(0.0, self.period - (time.time() - started))

    def hold(self, started):
        left = self.remaining(started)
        if left > 0.0:
            time.sleep(left)
        return left


class _StatusLine:
    """One self-overwriting console line."""

    def __init__(self, stream=None):
        self.stream = stream or sys.stdout
        self.width = 0

    def show(self, text):
        pad = max(0, self.width - len(text))
        self.stream.write("\r" + text + " " * pad)
        self.stream.flush()
        self.width = len(text)

    def close(self):
        if self.width:
            self.stream.write("\n")
            self.width = 0


def _clock(seconds):
    return str(timedelta(seconds=round(float(seconds), 2)))[:11]


class Scenario:
    def __init__(self, step_func, t_0=0, t_s=0.01, t_r=0.01, animate=False, axes=None, verbose=True,
                 t_snapshots=(), write_animation=False, dir_animation_out=None, fname_animation_out=None,
                 tempdir_animation=None, keep_animation_frames=False):
        if write_animation:
            raise NotImplementedError("writing animation videos is outside the scope of the MI355X engine")
        if not callable(step_func):
            raise TypeError("step_func must be callable")
        self.step_func = step_func
        self.t_0 = t_0
        self.t = t_0
        self.t_s = t_s
        self.t_r = t_r
        self.t_wall = time.time()
        self.i = 0
        self.i_end = 0
        self.animate = bool(animate)
        self.ax = axes
        self.verbose = verbose
        self.t_snapshots = tuple(t_snapshots)
        self.write_animation = False
        self.dir_animation_out = dir_animation_out
        self.fname_animation_out = fname_animation_out
        self.tempdir_animation = tempdir_animation
        self.keep_animation_frames = keep_animation_frames
        self.hist_run_time = []
        self.fig = None

    # ------------------------------------------------------------------ public
    def run(self, t_end):
        """Advance the scenario to simulated time t_end (scenario.py:96-113)."""
        if self.verbose:
            input("\nPress any key to start simulation ... \n")
        began = time.time()
        self.i_end = int(t_end / self.t_s)
        pacer = _RealTimePacer(self.t_r)
        status = _StatusLine() if self.verbose else None
        canvas = self._open_canvas() if self.animate else None
        while self.i < self.i_end:
            tick_began = time.time()
            if canvas is not None:
                self._step_blitting(canvas)
            else:
                self._step()
            self.hist_run_time.append(time.time() - tick_began)
            if status is not None:
                busy = time.time() - tick_began
                period = max(busy, self.t_r)
                rate = int(1.0 / period) if period > 0 else 0
                status.show(f"step {self.i} of {self.i_end} | simulated {_clock(self.t - self.t_0)} | "
                            f"elapsed {_clock(time.time() - began)} | {rate} ticks/s")
            pacer.hold(tick_began)
        if status is not None:
            status.close()
            print(f"Simulation finished after {_clock(time.time() - began)}")

    def reset(self):
        """Rewind the counters (scenario.py:226-228); the step function's own state is the caller's business."""
        self.i = 0
        self.t = self.t_0

    def ticks_per_second(self):
        """Mean tick rate of the ticks run so far, before pacing."""
        total = sum(self.hist_run_time)
        return len(self.hist_run_time) / total if total > 0 else float("inf")

    # ------------------------------------------------------------------ one tick
    def _step(self):
        """scenario.py:169-173"""
        self.step_func()
        self.i += 1
        self.t += self.t_s

    # ------------------------------------------------------------------ animation (matplotlib, optional)
    def _open_canvas(self):
        try:
            import matplotlib.pyplot as plt
        except ImportError as exc:  # pragma: no cover - matplotlib is an optional dependency
            raise RuntimeError("animate=True needs matplotlib") from exc
        if self.ax is None:
            self.fig, self.ax = plt.subplots(1, 1)
        else:
            self.fig = self.ax.figure
        self.ax.set_aspect("equal")
        self.fig.canvas.draw()
        return {"background": self.fig.canvas.copy_from_bbox(self.fig.bbox)}

    def _step_blitting(self, canvas):
        """Redraw only the animated artists of the vehicles' drawings on top of the saved background."""
        self.fig.canvas.restore_region(canvas["background"])
        self._step()
        for artist in self.ax.get_children():
            if getattr(artist, "get_animated", lambda: False)():
                self.ax.draw_artist(artist)
        self.fig.canvas.blit(self.fig.bbox)
        self.fig.canvas.flush_events()
