"""Vehicle classes with the constructors, attributes and methods of `cyclistsocialforce.vehicle`
(Vehicle :49, Bicycle :990, TwoDBicycle :1292, InvPendulumBicycle :1651, PlanarPointBicycle :1991),
executed by the HIP engine.

A vehicle owns NumPy state (`s`, `traj`, `trajF`, `destqueue`, ...) exactly like the reference object.
The numbers are produced on the GPU:

* inside a `SocialForceIntersection` the vehicle is one row of the population engine and `s` is refreshed
  from the device after every tick;
* stepped on its own (`calcDestinationForce`, `calcRepulsiveForce`, `step(Fx, Fy)` — the seam-1 hooks of
  SURVEY.md §8(b), used by calibration replay, calibration.py:438-460) it drives a private one-agent engine.

There is no NumPy fallback: without libcsf_hip.so / a GPU these methods raise `EngineError`.
"""
import numpy as np

from . import _ffi
from .engine import Engine
from .parameters import (BalancingRiderBicycleParameters, BicycleParameters, CarParameters, InvPendulumBicycleParameters,
                         PlanarBicycleParameters, PlanarPointBicycleParameters, VehicleParameters)
from .utils import limitAngle


class Vehicle:
    """Parent class — vehicle.py:49-917.  Abstract here: a concrete class selects the rider model."""

    PARAMS_TYPE = VehicleParameters
    MODEL = None
    N_STATES = 4
    STATE_NAMES = ["x[m]", "y[m]", "psi[rad]", "v[m/s]"]

    def __init__(self, s0, id="unknown", route=(), saveForces=False, params=None, dest_force_func=None,
                 rep_force_func=None):
        # per-vehicle hooks (vehicle.py:56, 194-204, 225-234): None selects the class's own force functions, which run
        # on the GPU.  A custom callable is the caller's Python code: honoured by the single-vehicle methods below, and inside
        # a SocialForceIntersection by a tick that forms the forces on the host from the engine's pieces (intersection.py of
        # this package: _hooked_forces) - meant for the handful of road users such scenarios have.
        self._owner = None
        self.dest_force_func = dest_force_func
        self.rep_force_func = rep_force_func
        if self.MODEL is None:
            raise NotImplementedError("instantiate Bicycle, TwoDBicycle, InvPendulumBicycle or PlanarPointBicycle")
        if params is None:                                             # vehicle.py:139-143
            self.params = self.PARAMS_TYPE()
        else:
            if not isinstance(params, self.PARAMS_TYPE):
                raise TypeError(f"Params must be a '{self.PARAMS_TYPE.__name__}' object. "
                                f"Instead it was '{type(params).__name__}'.")
            self.params = params
        self._owner = None        # SocialForceIntersection that holds this vehicle, or None
        self._index = -1
        self._live = False        # True once the intersection's engine holds this vehicle (bulk mirror active)
        self.i = 0                                                     # vehicle.py:146
        if len(s0) < self.N_STATES:                                    # vehicle.py:149-152
            raise ValueError(f"The initial state s0 has to be size {self.N_STATES} with states "
                             f"{self.STATE_NAMES}. Instead it was {s0}.")
        if len(s0) > self.N_STATES:
            s0 = s0[:self.N_STATES]
        self.s = np.array(s0, dtype=float)                             # vehicle.py:154-156
        self.s[2] = limitAngle(s0[2])
        self.s_names = list(self.STATE_NAMES)
        self.traj = np.zeros((len(s0), int(30 / self.params.t_s)))     # vehicle.py:159-160
        self.traj[:, 0] = self.s
        self.saveForces = saveForces
        if self.saveForces:
            self.trajF = np.zeros((2, int(30 / self.params.t_s)))
        assert isinstance(id, str), "User ID has to be a string."      # vehicle.py:167-168
        self.id = id
        assert isinstance(route, tuple), "Route has to be a tuple"     # vehicle.py:171-177
        assert all(isinstance(r, str) for r in route), "Edge IDs in route list have to be str"
        self.follow_route = bool(route)
        self.route = route
        self.drawing = None
        self.destqueue = np.c_[float(s0[0]), float(s0[1]), 0.0]       # vehicle.py:183-185: (x0, y0, no stop)
        self.destpointer = 0
        self.znav = np.array([True, False, False])                     # vehicle.py:188
        self.F = []
        self.force = (0.0, 0.0)
        self.uncontrolled = False
        # engine binding
        self._solo = None         # private one-agent engine
        self._solo_synced = False
        self._queue_dirty = True
        self._s_shadow = self.s.copy()

    # hooks may be assigned at any time (the reference's are plain attributes): the intersection that holds the vehicle is told
    @property
    def rep_force_func(self):
        return self._rep_force_func

    @rep_force_func.setter
    def rep_force_func(self, f):
        self._rep_force_func = f
        if f is not None and self._owner is not None:
            self._owner._hooked = True

    @property
    def dest_force_func(self):
        return self._dest_force_func

    @dest_force_func.setter
    def dest_force_func(self, f):
        self._dest_force_func = f
        if f is not None and self._owner is not None:
            self._owner._hooked = True

    # ------------------------------------------------------------------ mirrored scalars
    # Inside a SocialForceIntersection these live in the intersection's bulk arrays (one device read-back per tick
    # refreshes every vehicle at once); on its own the vehicle keeps them itself.
    @property
    def s(self):
        """the state vector (vehicle.py:154-156).  Inside an intersection it is a row view of the bulk mirror, and whoever
        holds it may write through it (calibration.py:455-460 edits vehicle.s in place): the intersection is told, and
        compares its mirror with what the device holds before the next tick - only from then on, not on every tick of a
        population nobody looks at"""
        if self._live:
            self._owner._state_handed_out()
        return self._s

    @s.setter
    def s(self, value):
        if self._live:                                                 # rebinding vehicle.s: the mirror's row takes the values
            row = self._s
            if value is not row:
                self._owner._state_handed_out()
                row[:] = np.asarray(value, dtype=float)[: row.size]
        else:
            self._s = value

    @property
    def i(self):
        """column of `traj` written last (vehicle.py:146, 1279-1282)"""
        return int(self._owner._ti[self._index]) if self._live else self._i

    @i.setter
    def i(self, value):
        self._i = int(value)
        if self._live:
            self._owner._ti[self._index] = int(value)

    @property
    def destpointer(self):
        return int(self._owner._ptr[self._index]) if self._live else self._destpointer

    @destpointer.setter
    def destpointer(self, value):
        self._destpointer = int(value)
        if self._live:
            self._owner._ptr[self._index] = int(value)

    @property
    def dest(self):
        """current destination (x, y, stop): the row of the queue the pointer is on (vehicle.py:183-185, 545-594)"""
        return self.destqueue[self.destpointer, :]

    @dest.setter
    def dest(self, value):  # the reference rebinds `dest` to the queue row; here it always is that row
        pass

    @property
    def force(self):
        """total force of the last evaluated tick (intersection.py:860-861)"""
        if self._live and self._owner._have_force:
            return (float(self._owner._fx[self._index]), float(self._owner._fy[self._index]))
        return self._force

    @force.setter
    def force(self, value):
        self._force = (float(value[0]), float(value[1]))
        if self._live:
            self._owner._fx[self._index], self._owner._fy[self._index] = self._force

    @property
    def drawing(self):
        return self._drawing

    @drawing.setter
    def drawing(self, value):
        self._drawing = value
        if self._owner is not None:
            self._owner._drawn_stale = True

    @property
    def F(self):
        """magnitudes of the total force, one entry per evaluated tick (intersection.py:862)"""
        if self._owner is not None:
            self._owner._fold_force_log(self)
        return self._F

    @F.setter
    def F(self, value):
        self._F = list(value)

    # ------------------------------------------------------------------ engine plumbing
    def _pod(self, priority_rule=0):
        return self.params.to_pod(self.MODEL, priority_rule)

    def _solo_engine(self):
        """The vehicle's own one-agent engine, kept in step with the Python-side attributes."""
        if self._owner is not None:
            raise RuntimeError("this vehicle belongs to a SocialForceIntersection; step the intersection")
        if self._solo is None:
            self._solo = Engine(self._pod(), 1)
            self._solo.add_agents(self.s[None, :], float(self.params.v_desired_default))
            self._queue_dirty = True
            self._s_shadow = self.s.copy()
            self._vd_shadow = float(self.params.v_desired_default)
        e = self._solo
        if self._queue_dirty:
            keep = getattr(self, "_queue_keep_ptr", False) and self._solo_synced
            e.set_dest_queue([0], [0, self.destqueue.shape[0]], self.destqueue, reset=2 if keep else 1)
            self._queue_dirty = False
            self._queue_keep_ptr = False
            self._solo_synced = True
        if not np.array_equal(self.s, self._s_shadow):
            e.push_state([0], self.s[None, :])
            self._s_shadow = self.s.copy()
        if float(self.params.v_desired_default) != self._vd_shadow:
            e.set_v_desired([0], float(self.params.v_desired_default))
            self._vd_shadow = float(self.params.v_desired_default)
        return e

    def _pull(self, s_row, ptr, znav_row):
        """Adopt the engine's view after a device tick (also advances the traj ring like vehicle.step)."""
        self.s[:] = s_row
        self._s_shadow[:] = s_row
        self.destpointer = int(ptr)
        self.znav[:] = znav_row

    def _advance_history(self, Fx, Fy):
        self.i = (self.i + 1) % self.traj.shape[1]                     # vehicle.py:1279-1282 (see DESIGN D5)
        self.traj[:, self.i] = self.s
        if self.saveForces:
            self.trajF[0, self.i] = Fx
            self.trajF[1, self.i] = Fy
        self.update_drawing(Fres=(Fx, Fy))

    # ------------------------------------------------------------------ reference API
    def calcRepulsiveForce(self, x, y, psi):
        """vehicle.py:250-279 / 1560-1648: force this vehicle exerts on road users at (x, y, psi)."""
        if self.rep_force_func is not None:
            return self.rep_force_func(self, x, y, psi)
        return self.class_field(x, y, psi)

    def class_field(self, x, y, psi):
        """The field of the vehicle's CLASS whatever hook it carries - what the reference's classes reach by calling
        `TwoDBicycle.calcRepulsiveForce(vehicle, x, y, psi)` unbound (vehicle.py:982, 1987, 2024, 2069): a hook that scales or
        reshapes its class's own field calls this."""
        if getattr(self.params, "f_0", 1.0) == 0.0 and self.MODEL != _ffi.BICYCLE:
            return 0.0, 0.0                                            # vehicle.py:1592-1593
        x = np.atleast_1d(np.asarray(x, dtype=float)).ravel()
        y = np.atleast_1d(np.asarray(y, dtype=float)).ravel()
        psi = np.atleast_1d(np.asarray(psi, dtype=float)).ravel() if psi is not None else np.zeros_like(x)
        eng = self._owner._engine_ready() if self._owner is not None else self._solo_engine()
        return eng.pair_force(np.r_[self._s[:3], self._s[3]], x, y, psi)

    def calcDestinationForce(self):
        """vehicle.py:281-299, 1189-1194, 1416-1558 (advances the destination queue / nav state)."""
        if self.dest_force_func is not None:                           # vehicle.py:295-297
            self.updateDestination()
            return self.dest_force_func(self)
        e = self._solo_engine()
        fx, fy = e.dest_force()
        s, ptr, zn, _ = e.state(with_nav=True)
        self.destpointer = int(ptr[0])
        self.znav[:] = zn[0]
        return float(fx[0]), float(fy[0])

    def step(self, F1=0, F2=0):
        """vehicle.py:301-328, 1274-1289, 1386-1414, 1883-1930: one control + kinematics step."""
        e = self._solo_engine()
        e.apply_forces([float(F1)], [float(F2)])
        s, ptr, zn, _ = e.state(with_nav=True)
        self._pull(s[0], ptr[0], zn[0])
        self._advance_history(float(F1), float(F2))

    def _engine_and_row(self):
        """(engine, row) that hold this vehicle on the device, with every host-side edit pushed"""
        if self._owner is not None:
            return self._owner._push_mutations(), self._index
        return self._solo_engine(), 0

    def _refresh_nav(self, e, row):
        _, ptr, zn, _ = e.state(with_nav=True)
        self.destpointer = int(ptr[row])
        self.znav[:] = zn[row]

    def updateDestination(self):
        """vehicle.py:545-594: advance the queue pointer past destinations that were reached or can be skipped."""
        assert self.destqueue is not None, "Road user does not have a destination queue!"
        e, row = self._engine_and_row()
        e.update_destination([row])
        self._refresh_nav(e, row)

    def updateNavState(self, stop):
        """vehicle.py:354-457: one transition of the cruise / decelerate / arrived machine; returns (vd, ddest)."""
        assert np.any(self.znav), "Invalid state!"
        e, row = self._engine_and_row()
        vd, ddest = e.update_nav_state([row], [1 if bool(stop) else 0])
        self._refresh_nav(e, row)
        return float(vd[0]), float(ddest[0])

    def getDestinationDistance(self):
        """vehicle.py:596-604"""
        dest = self.destqueue[self.destpointer, :]
        return np.sqrt(np.power(dest[0] - self._s[0], 2) + np.power(dest[1] - self._s[1], 2))

    def isLastDest(self):
        """vehicle.py:537-543"""
        if self.destqueue is None:
            return True
        return self.destpointer + 1 >= np.shape(self.destqueue)[0]

    def setDestinations(self, x, y, stop=None, reset=False):
        """vehicle.py:606-647"""
        x = np.array([x], dtype=float).flatten()
        y = np.array([y], dtype=float).flatten()
        stop = np.zeros_like(x) if stop is None else np.array([stop], dtype=float).flatten()
        if reset or self.destqueue is None:
            self.destqueue = np.c_[x, y, stop]
            self.destpointer = 0
        else:
            self.destqueue = np.vstack((self.destqueue, np.c_[x, y, stop]))
        self._queue_dirty = True
        if self._owner is not None:
            self._owner._mark_queue_dirty(self, -1 if reset else None)

    def stop(self, stoptype=0, stopdest=None):
        """vehicle.py:459-503.  Type 0 sets the stop flag of the current destination (the reference does it through
        `self.dest`, a view of the queue row).  Type 2 rebinds `self.dest` to a detached (xstop, ystop, 1) array, which the
        next updateDestination() replaces by the queue row again, and steps the pointer back: only the pointer change
        outlives the call, and that is what is mirrored.  Type 1 reads `params.AMAX`, which no parameter class defines
        (vehicle.py:486): AttributeError, as in the reference."""
        if stoptype == 0:
            self.destqueue[self.destpointer, 2] = 1.0
            self._queue_edit()
        elif stoptype == 1:
            raise AttributeError(f"'{type(self.params).__name__}' object has no attribute 'AMAX'")
        elif stoptype == 2:
            if stopdest is None or len(stopdest) < 2:
                raise TypeError("stoptype 2 needs stopdest = (xstop, ystop)")
            if self.destpointer > 0:
                self._set_destpointer(self.destpointer - 1)
        else:
            raise ValueError("Stop type has to be one of [0,1,2].")

    def go(self, gotype=0):
        """vehicle.py:505-535: type 0 clears the stop flag of the current destination, type 1 re-reads the current queue
        row; both end with updateDestination()."""
        if gotype == 0:
            self.destqueue[self.destpointer, 2] = 0.0
            self._queue_edit()
        self.arrived = False
        self.updateDestination()

    def _set_destpointer(self, value):
        e, row = self._engine_and_row()
        e.set_dest_pointer([row], [int(value)])
        self.destpointer = int(value)

    def _queue_edit(self):
        """Rows were edited in place: the engine's copy is replaced, the destination pointer kept."""
        self._queue_dirty = True
        self._queue_keep_ptr = True
        if self._owner is not None:
            self._owner._mark_queue_dirty(self, -2)

    def setSplineDestinations(self, x, y, npoints, stop=False, reset=False):
        """vehicle.py:649-693 (host-side convenience; uses scipy like the reference)."""
        from scipy import interpolate

        assert len(x) >= 3, "Provide at least 3 points to calculate a cubic trajectory prototype"
        x = np.insert(np.array(x, dtype=float), 0, self._s[0])
        y = np.insert(np.array(y, dtype=float), 0, self._s[1])
        tck, _ = interpolate.splprep((x, y), s=0.0)
        x_i, y_i = interpolate.splev(np.linspace(0, 1, npoints), tck)
        if stop:
            flags = np.zeros_like(x_i)
            flags[-1] = 1.0
            self.setDestinations(x_i, y_i, stop=flags, reset=reset)
        else:
            self.setDestinations(x_i, y_i, reset=reset)

    def add_drawing(self, ax, drawing=None, **kwargs):
        """vehicle.py:695-720: attach a drawing to this vehicle - the one passed in (anything with
        update(vehicle, Fdest=, Frep=, Fres=) and set_animated(flag)), else a `vizualisation.VehicleDrawing` on `ax`."""
        if drawing is None:
            from .vizualisation import VehicleDrawing
            drawing = VehicleDrawing(ax, self, **kwargs)
        self.drawing = drawing

    def plot_states(self, t_end=None, axes=None):
        """vehicle.py:734-787"""
        from .vizualisation import plot_states
        return plot_states(self, t_end=t_end, axes=axes)

    def plot_forces(self, t_end=None, axes=None, components_to_plot=("magnitude", "direction")):
        """vehicle.py:789-917"""
        from .vizualisation import plot_forces
        return plot_forces(self, t_end=t_end, axes=axes, components_to_plot=components_to_plot)

    def update_drawing(self, Fdest=None, Frep=None, Fres=None):
        """vehicle.py:722-732"""
        if self.drawing is not None:
            self.drawing.update(self, Fdest=Fdest, Frep=Frep, Fres=Fres)


class Bicycle(Vehicle):
    """vehicle.py:990-1289 — 2D kinematic bicycle, elliptic repulsive field, straight-line destination force."""

    PARAMS_TYPE = BicycleParameters
    MODEL = _ffi.BICYCLE
    N_STATES = 5
    STATE_NAMES = ["x[m]", "y[m]", "psi[rad]", "v[m/s]", "delta[rad]"]

    def __init__(self, s0, **kwargs):
        Vehicle.__init__(self, s0, **kwargs)
        self.destspline = None


class TwoDBicycle(Bicycle):
    """vehicle.py:1292-1648 — "2D model": spline path planner + the new repulsive field.

    The reference constructor is broken at HEAD (vehicle.py:1359, SURVEY finding 2); the documented
    signature is honoured here."""

    PARAMS_TYPE = InvPendulumBicycleParameters
    MODEL = _ffi.TWOD

    def __init__(self, s0, id="unknown", route=(), saveForces=False, params=None, **hooks):
        Bicycle.__init__(self, s0, id=id, route=route, saveForces=saveForces, params=params, **hooks)


class InvPendulumBicycle(TwoDBicycle):
    """vehicle.py:1651-1950 — inverted-pendulum roll/steer/yaw loop."""

    MODEL = _ffi.INVPEND
    N_STATES = 6
    STATE_NAMES = ["x[m]", "y[m]", "psi[rad]", "v[m/s]", "delta[rad]", "theta[rad]"]

    def __init__(self, s0, **kwargs):
        TwoDBicycle.__init__(self, s0, **kwargs)


InvertedPendulumBicycle = InvPendulumBicycle  # README.md:14 name


class PlanarPointBicycle(Vehicle):
    """vehicle.py:1991-2028 — mass-less particle with first-order yaw tracking (dynamics.py:802-1079)."""

    PARAMS_TYPE = PlanarPointBicycleParameters
    MODEL = _ffi.PLANARPOINT

    def __init__(self, s0, **kwargs):
        Vehicle.__init__(self, s0, **kwargs)


class PlanarBicycle(Vehicle):
    """vehicle.py:2031-2076 — planar two-wheeler: pole-placed steer / yaw loop whose gains follow the speed every step
    (dynamics.py:178-258, 1167-1226), first-order speed dynamics (dynamics.py:145-175), the TwoD force field and the
    spline destination force."""

    PARAMS_TYPE = PlanarBicycleParameters
    MODEL = _ffi.PLANARBIKE
    N_STATES = 5
    STATE_NAMES = ["x[m]", "y[m]", "psi[rad]", "v[m/s]", "delta[rad]"]

    def __init__(self, s0, **kwargs):
        assert len(s0) >= 5, ("s0 has to have at least five elements:", " (x, y, psi, v, delta)!")
        Vehicle.__init__(self, s0, **kwargs)


class UncontrolledVehicle(Vehicle):
    """vehicle.py:920-988 — a stationary or externally controlled vehicle (a car): it follows the states prescribed in its
    `traj` (column i at tick i, while there is one), exerts the TwoDBicycle force field with its CarParameters and feels
    no force.  Inside a SocialForceIntersection the prescribed trajectory is handed to the engine when the vehicle is
    attached (csf_set_script); `traj` stays the vehicle's own array."""

    PARAMS_TYPE = CarParameters
    MODEL = _ffi.UNCONTROLLED

    def __init__(self, s0, trajectory=(), **kwargs):
        Vehicle.__init__(self, s0, **kwargs)
        self.uncontrolled = True
        self._script = None
        if len(trajectory) > 0:                                        # vehicle.py:958-960
            self.traj = np.array(trajectory, dtype=float)
            if self.traj.ndim != 2 or self.traj.shape[0] != self.N_STATES:
                raise ValueError(f"trajectory must be [{self.N_STATES}, T]: one column of (x, y, psi, v) per tick")
            self._script = np.ascontiguousarray(self.traj.T)

    def calcDestinationForce(self):                                    # vehicle.py:987-988
        return 0, 0


class BalancingRiderBicycle(Vehicle):
    """vehicle.py:1953-1990 — a bicycle with Whipple-Carvallo dynamics under the rider's full-state feedback
    (dynamics.py:261-705): eight states (x, y, psi, v, delta, phi, deltadot, phidot), gains placed every step for the poles the
    control model asks for at the current speed, implicit midpoint rule; the TwoDBicycle force field and the direct-approach
    destination force."""

    PARAMS_TYPE = BalancingRiderBicycleParameters
    MODEL = _ffi.BALANCINGRIDER
    N_STATES = 8
    STATE_NAMES = ["x[m]", "y[m]", "psi[rad]", "v[m/s]", "delta[rad]", "phi[rad]", "deltadot[rad/s]", "phidot[rad/s]"]

    def __init__(self, s0, **kwargs):
        Vehicle.__init__(self, s0, **kwargs)
        # dynamics.py:306: the rider's first gains are placed at the start speed - with stochastic_control_behavior that is the first
        # draw from the pole model (in the order the vehicles are made, on NumPy's global generator: parameters.py:1391-1396)
        if getattr(self.params, "stochastic_control_behavior", False):
            self.params.update_control_params(float(self._s[3]))
