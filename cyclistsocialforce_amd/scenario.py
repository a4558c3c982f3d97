"""Tick driver with the constructor and `run()` semantics of `cyclistsocialforce.scenario.Scenario`
(scenario.py:53-265), free of the reference module's import-time dependencies (sumolib, traci, cv2,
mypyutils: scenario.py:13-50).  Animation and video write-out are out of scope; everything else — the
real-time pacing of `_wait`, the progress line, the `input()` prompt when verbose — behaves the same, so
benchmarks must pass `t_r=0, verbose=False` exactly as they would with the reference.
"""
from datetime import timedelta
from time import sleep, time


class Scenario:
    def __init__(self, step_func, t_0=0, t_s=0.01, t_r=0.01, animate=False, axes=None, verbose=True,
                 t_snapshots=(), write_animation=False, dir_animation_out=None, fname_animation_out=None,
                 tempdir_animation=None, keep_animation_frames=False):
        if animate or write_animation:
            raise NotImplementedError("animation is outside the scope of the MI355X engine")
        self.t = t_0                                                   # scenario.py:75-94
        self.t_s = t_s
        self.t_r = t_r
        self.t_0 = t_0
        self.t_wall = time()
        self.i = 0
        self.animate = False
        self.ax = axes
        self.verbose = verbose
        self.step_func = step_func
        self.hist_run_time = []                                        # per-tick wall time (scenario.py:301)

    def run(self, t_end):
        """scenario.py:96-113"""
        if self.verbose:
            input("\nPress any key to start simulation ... \n")
        t_start = time()
        self._run_silent(t_start, t_end)
        if self.verbose:
            print("\n")
            print(f"Simulation finished after {str(timedelta(seconds=time() - t_start))[:-3]}")

    def _run_silent(self, t_start, t_end):
        """scenario.py:115-122"""
        self.i_end = int(t_end / self.t_s)
        len_prev_msg = 0
        while self.i < self.i_end:
            t = time()
            self._step()
            self.hist_run_time.append(time() - t)
            len_prev_msg = self._wait(t, t_start, self.i_end, len_prev_msg)

    def _step(self):
        """scenario.py:169-173"""
        self.step_func()
        self.i += 1
        self.t += self.t_s

    def _wait(self, t, t_start, i_end, len_prev_msg):
        """scenario.py:175-195: sleep up to the real-time step t_r and print the progress line."""
        dt = time() - t
        t_sleep = max(0, self.t_r - dt)
        msg = ""
        if self.verbose:
            sim_time = str(timedelta(seconds=self.t))[:11]
            wall_time = str(timedelta(seconds=(time() - t_start)))[:11]
            msg = (f"Running step {self.i}/{i_end}, Sim. time {sim_time}, Wall time {wall_time}, "
                   f"Wall freq. {int(1 / (dt + t_sleep)) if dt + t_sleep > 0 else 0} Hz ")
            msg += " " * max(len_prev_msg - len(msg), 0)
            print("\r" + msg, end="")
        if dt < self.t_r:
            sleep(t_sleep)
        return len(msg)

    def reset(self):
        """scenario.py:226-228"""
        self.i = 0
        self.t = self.t_0
