"""`SocialForceIntersection` and the road-element classes with the interface of
`cyclistsocialforce.intersection` (SocialForceIntersection :253-916, RoadEdge :214, RoadSegment :72,
Straight/CurvedRoadSegment :118/:149, RoadSegmentCollection :32), driving the HIP engine.

One `step()` is exactly one `csf_step(engine, 1)`: FOV mask, all-pairs repulsive field, clamp, destination
force, road-edge force, controller + kinematics for every vehicle, snapshot refresh — all on the GPU.
The Python objects are a mirror: user mutations of `vehicle.s`, `vehicle.params.v_desired_default`, the
destination queues or the population are pushed to the device before the next tick, and `vehicle.s`,
`traj`, `destpointer`, `znav`, `force`, `F` are refreshed after it.  `step_n(k)` runs k ticks without any
per-tick Python work.

SUMO co-simulation: with `activate_sumo_cosimulation=True` the intersection reads its footprint,
approach lanes and internal lanes from a sumolib-style `net`, finds arrivals and departures on its internal lanes
through TraCI, gives arriving road users a spline prototype across the junction, and pushes every position back with
`traci.vehicle.moveToXY` after each tick (intersection.py:341-453, 458-539, 679-688).  `traci` is imported on demand
or injected (`traci=`), so the seam is testable without a SUMO installation.  Arrivals and departures reach the device
incrementally (include/csf.h: csf_set_incremental).

`animate=True` gives every vehicle a `vizualisation.VehicleDrawing` on the axes passed in (intersection.py:881-885); the
drawings are refreshed from the host mirror after each tick's read-back.

Out of scope: populations mixing vehicle classes in one intersection, writing animation videos.
"""
import numpy as np

from . import _ffi, parameters
from .engine import Engine
from .parameters import RoadElementParameters
from .utils import angleSFMtoSUMO, generateSplinePrototype
from .vehicle import Vehicle

PRIORITY_RULES = {"unregulated": _ffi.UNREGULATED, "p2r": _ffi.P2R}


# ----------------------------------------------------------------------------- road elements

class RoadEdge:
    """intersection.py:214-250: a polyline whose vertices repel road users with -F_0 r^-sigma."""

    def __init__(self, vertices, params=None):
        self.vertices = np.asarray(vertices, dtype=float)
        self.params = params if params is not None else RoadElementParameters()

    def edges(self):
        return [self]


class RoadSegment:
    """intersection.py:72-115: two edges at +-width/2 around a centre line starting at x0 = (x, y, heading).
    As in the reference the segment's own `params` are defaults; the edges carry the ones passed in."""

    def __init__(self, x0, width, ds=0.1, params=None):
        self.params = RoadElementParameters()
        self.x0 = x0
        self.x1 = x0
        self.width = width
        self.edge_list = []
        self.ds = ds

    def edges(self):
        return list(self.edge_list)

    @staticmethod
    def _place(local_xy, origin, heading):
        c, s = np.cos(heading), np.sin(heading)
        rot = np.array([[c, -s], [s, c]])
        return (rot @ np.asarray(local_xy, dtype=float).T).T + np.asarray(origin[:2], dtype=float)


class StraightRoadSegment(RoadSegment):
    """intersection.py:118-146: vertices every `ds` along a straight of `length`."""

    def __init__(self, x0, width, length, ds=0.1, params=None):
        RoadSegment.__init__(self, x0, width, ds, params)
        params = params if params is not None else RoadElementParameters()
        self.length = length
        along = np.arange(0, length + self.ds, self.ds)                # :126
        for side in (-0.5, 0.5):                                       # right edge first (:127-140)
            local = np.c_[along, np.full_like(along, side * width)]
            self.edge_list.append(RoadEdge(self._place(local, x0, x0[2]), params=params))
        self.x1 = np.zeros_like(np.asarray(x0, dtype=float))           # :142-146
        self.x1[:2] = np.asarray(x0[:2], dtype=float) + length * np.array([np.cos(x0[2]), np.sin(x0[2])])
        self.x1[2] = x0[2]


class CurvedRoadSegment(RoadSegment):
    """intersection.py:149-211: circular arc of `radius` through `angle`, turning "left" or "right"."""

    def __init__(self, x0, width, radius, angle, direction, ds=0.1, params=None):
        RoadSegment.__init__(self, x0, width, ds, params)
        params = params if params is not None else RoadElementParameters()
        assert direction in ("left", "right"), f'direction has to be "left" or "right, instead it was {direction}'
        self.length = radius * angle
        self.radius, self.angle, self.direction = radius, angle, direction
        turn = 1 if direction == "left" else -1                        # :166
        beta = x0[2] - np.pi / 2                                       # :173
        for sign in (+1, -1):                                          # right edge first (:176-207)
            r_edge = radius + sign * turn * width / 2
            grid = np.linspace(0, angle, int(r_edge * angle / self.ds))
            local = np.c_[turn * (r_edge * np.cos(grid) - radius), r_edge * np.sin(grid)]
            self.edge_list.append(RoadEdge(self._place(local, x0, beta), params=params))
        end_local = np.array([[turn * (radius * np.cos(angle) - radius), radius * np.sin(angle)]])
        self.x1 = np.zeros(3)                                          # :209-211
        self.x1[:2] = self._place(end_local, x0, beta)[0]
        self.x1[2] = x0[2] + turn * angle


class RoadSegmentCollection:
    """intersection.py:32-69"""

    def __init__(self, segs):
        self.segs = segs

    def edges(self):
        return [e for seg in self.segs for e in seg.edges()]

    def get_destinations_from_segments(self):
        return [seg.x1[0] for seg in self.segs], [seg.x1[1] for seg in self.segs]

    def __getitem__(self, i):
        if not isinstance(i, int):
            raise ValueError("Subscription index must be integer!")
        if i > len(self.segs):
            raise IndexError(f"RoadSegmentCollection has {len(self.segs)} segments, but i={i} was requested")
        return self.segs[i]


def flatten_road_elements(elements):
    """(offsets[n_edges+1], vertices[sum,2], F0[n_edges], sigma[n_edges]) of a road_elements list."""
    edges = [e for el in elements for e in el.edges()]
    off = np.zeros(len(edges) + 1, dtype=np.int64)
    for k, e in enumerate(edges):
        off[k + 1] = off[k] + e.vertices.shape[0]
    verts = np.vstack([e.vertices for e in edges]) if edges else np.zeros((0, 2))
    return off, verts, np.array([e.params.F_0 for e in edges]), np.array([e.params.sigma for e in edges])


# ----------------------------------------------------------------------------- the population tick

class SocialForceIntersection:
    """Open space / intersection managing one population of road users (intersection.py:253-916)."""

    def __init__(self, vehicleList, id="", priority_rule="unregulated", animate=False, axes=None,
                 activate_sumo_cosimulation=False, net=None, road_elements=[], bicycle_drawing_kwargs={},
                 capacity=None, device=0, track_params=True, traci=None):
        if animate and axes is None:
            raise AssertionError("Provide axes for animation!")        # intersection.py:409
        if priority_rule not in PRIORITY_RULES:
            raise ValueError(f"priority_rule must be one of {tuple(PRIORITY_RULES)}")
        assert isinstance(id, str), "Intersection ID has to be a string."
        self.bicycle_drawing_kwargs = bicycle_drawing_kwargs
        self.is_first_step = True
        self.activate_sumo_cosimulation = bool(activate_sumo_cosimulation)
        self._traci = traci
        self.id = id
        self.priority_rule = priority_rule
        self.animate = bool(animate)
        self.ax = axes
        self.road_elements = road_elements
        self.hist_n_vecs = []
        self.vehicles = []
        self.n_bikes = 0
        self._vx = np.zeros((0, 1))       # vehicleX / vehicleY / vehicleTheta (intersection.py:660-677): refreshed when read
        self._vy = np.zeros((0, 1))
        self._vth = np.zeros((0, 1))
        self._pos_stale = False
        self._device = device
        self._capacity = capacity
        self._track_params = track_params
        self._engine = None
        self._S = None            # host mirror [capacity, n_states]; vehicle.s are row views
        self._shadow = None       # device's view of the mirror after the last pull / push, kept once a vehicle.s has been handed out
        self._s_watched = False   # some vehicle.s has been handed out: the mirror is compared with the shadow before every tick
        self._scripted = []       # live UncontrolledVehicles (their traj may be rewritten while they run)
        self._vd = None
        self._rule_on_device = None
        self._pod_bytes = None
        self._class_table = None
        self._cls = None
        self._road_sig = None
        self._pending = []        # vehicles waiting to be added to the engine
        # bulk mirror of the per-vehicle attributes the device produces (Vehicle properties read these): one
        # read-back per tick refreshes every vehicle, with no Python loop over the population
        self._ptr = None          # destpointer [capacity]
        self._zn = None           # znav [capacity, 3]; vehicle.znav are row views
        self._fx = self._fy = None
        self._have_force = False
        self._hooked = False      # some vehicle carries a custom rep_force_func / dest_force_func: forces are formed on the host (_hooked_forces)
        self._field_engines = {}  # ... engines for the class fields of the others, by parameter set
        self._stoch, self._stoch_for, self._stoch_epoch = [], -1, 0   # riders with stochastic_control_behavior (cache)
        self._ti = None           # vehicle.i [capacity]
        self._traj = None         # vehicle.traj, stored [traj_len, rows, n_states] (a tick writes one contiguous slab);
                                  # vehicle.traj are transposed views [n_states, traj_len]
        self._dirty_queues = {}   # vehicle -> None (rows appended) | -1 (queue replaced) | -2 (rows edited in place)
        self._drawn = []          # vehicles with a drawing / saveForces (the only per-tick Python loop)
        self._drawn_stale = True
        self._flog = []           # |F| of every evaluated tick since the last fold (vehicle.F)
        self._flog_base = 0
        self._params_seen = -1
        if self.activate_sumo_cosimulation:
            self._init_sumo(net)
        for v in vehicleList:                                  # (as given: the reference plans no route for them, :316-317)
            self._attach(v)
        if self.activate_sumo_cosimulation:
            self.update_road_user_positions()                  # intersection.py:320

    # ------------------------------------------------------------------ SUMO co-simulation
    def _init_sumo(self, net):
        """intersection.py:341-405: footprint, the last / first two points of every approach / exit lane (sampled from
        a spline through the lane shape), and the internal lanes on which SUMO hands road users over."""
        from scipy import interpolate

        if net is None:
            raise ValueError("activate_sumo_cosimulation=True needs the sumolib net of the scenario (net=...)")
        if self._traci is None:
            try:
                import traci as _traci
            except ImportError as exc:
                raise ImportError("SUMO co-simulation needs the `traci` package (or pass traci=...)") from exc
            self._traci = _traci
        self.node = net.getNode(self.id)
        self.shape_vertices = np.asarray(self.node.getShape(), dtype=float)
        try:
            from matplotlib import path as pth
            self.shape = pth.Path(self.shape_vertices, closed=True)
        except ImportError:                                   # pragma: no cover - matplotlib is optional
            self.shape = self.shape_vertices

        def lane_ends(edges, order, take):
            ends = {}
            for edge in edges:
                ends[edge.getID()] = []
                for lane in edge.getLanes():
                    path = np.asarray(lane.getShape(), dtype=float)
                    tck, _ = interpolate.splprep((path[:, 0], path[:, 1]), s=0.0, k=min(order, path.shape[0] - 1))
                    x_i, y_i = interpolate.splev(np.linspace(0, 1, 10), tck)
                    ends[edge.getID()].append((x_i[take], y_i[take]))
            return ends

        self.inEdges = lane_ends(self.node.getIncoming(), 5, slice(-2, None))
        self.outEdges = lane_ends(self.node.getOutgoing(), 3, slice(None, 2))
        self.internal_lanes, self.internal_lane_ids = [], []
        for edge in net.getEdges():
            if edge.getFromNode() == self.node and edge.getToNode() == self.node:
                for lane in edge.getLanes():
                    self.internal_lane_ids.append(lane.getID())
                    self.internal_lanes.append(lane)
        if not self.internal_lanes:
            raise ValueError(f"Intersection {self.id} does not have internal lanes! Cyclistsocialforce requires internal "
                             "lanes to allocate SUMO road users to intersections. Check if the net-file correctly "
                             "specifies interal lanes for this intersection!")

    def find_entered_exited_roadusers(self):
        """intersection.py:429-453: ids that appeared on / vanished from the internal lanes during the last SUMO step."""
        before = self.get_road_user_ids()
        now = []
        for lane in self.internal_lane_ids:
            now += list(self._traci.lane.getLastStepVehicleIDs(lane))
        return np.setdiff1d(now, before), np.setdiff1d(before, now)

    def _route_prototype(self, user):
        """intersection.py:471-519: destinations across the junction for a road user that follows a SUMO route."""
        ecurrent, enext = user.route[0], user.route[1]
        assert ecurrent in self.inEdges, f"Road user {user.id} arriving on junction {self.id} from unknown edge {ecurrent}!"
        assert enext in self.outEdges, f"Road user {user.id} requesting to depart junction {self.id} on unknown edge {enext}!"
        lanes_in = self.inEdges[ecurrent]
        lane_in = 0
        if len(lanes_in) > 1:                                  # the closer of the first two approach lanes
            xs = np.concatenate((lanes_in[0][0], lanes_in[1][0]))
            ys = np.concatenate((lanes_in[0][1], lanes_in[1][1]))
            lane_in = int(np.argmin(np.hypot(xs - user._s[0], ys - user._s[1])) / 2)
        lane_out = np.random.randint(0, len(self.outEdges[enext]))
        pts = np.vstack((np.array(lanes_in[lane_in]).T, np.array(self.outEdges[enext][lane_out]).T))
        xp, yp = generateSplinePrototype(pts[:, 0], pts[:, 1], 5)
        ahead = np.hypot(xp - xp[-1], yp - yp[-1]) < np.hypot(user._s[0] - xp[-1], user._s[1] - yp[-1])
        user.setDestinations(xp[ahead], yp[ahead], reset=True)  # only the points still in front of the road user

    # ------------------------------------------------------------------ population management
    def _attach(self, v):
        if not isinstance(v, Vehicle):
            raise TypeError("road users must be cyclistsocialforce_amd.vehicle.Vehicle objects")
        if v._owner is not None and v._owner is not self:
            raise RuntimeError(f"vehicle {v.id} already belongs to another intersection")
        self._stoch_for = -1
        if v.dest_force_func is not None or v.rep_force_func is not None:
            self._hooked = True                                        # (this population's forces are formed on the host: _hooked_forces)
        if v._solo is not None:
            v._solo.close()
            v._solo = None
        v._owner = self
        v._index = len(self.vehicles)
        v._live = False
        v._queue_synced = -1
        v._f_seen = self._flog_base + len(self._flog)
        self._dirty_queues[v] = -1
        self.vehicles.append(v)
        self._pending.append(v)
        self.n_bikes = len(self.vehicles)
        self._pos_stale = True                                         # intersection.py:531-538: vehicleX ... grow by a row

    def _param_classes(self, vehicles=None):
        """The parameter sets of the population.  Every reference vehicle owns its params object: the field vehicle i
        exerts is evaluated with ITS f_0 / sigma / e (vehicle.py:1592-1612) and masked with ITS hfov
        (intersection.py:733-735), and it moves with its own gains and limits.  The engine holds a table of up to 256
        distinct csf_params and a row index per road user (csf_set_param_classes); `params.v_desired_default` is per
        road user anyway.  Returns (list of PODs, vehicle 0's first; int32 row of every vehicle)."""
        vehicles = self.vehicles if vehicles is None else vehicles
        rule = PRIORITY_RULES.get(self.priority_rule, 0)
        by_object, rows, pods = {}, {}, []
        cls = np.zeros(len(vehicles), dtype=np.int32)
        for k, v in enumerate(vehicles):
            c = by_object.get(id(v.params))
            if c is None:
                pod = v._pod(rule)
                key = bytes(pod)
                c = rows.get(key)
                if c is None:
                    if pods and (pod.t_s != pods[0].t_s or pod.traj_len != pods[0].traj_len):
                        raise NotImplementedError(f"vehicle {v.id!r}: t_s = {pod.t_s} but vehicle {vehicles[0].id!r} has "
                                                  f"{pods[0].t_s}; the road users of one intersection share the clock")
                    c = rows[key] = len(pods)
                    pods.append(pod)
                by_object[id(v.params)] = c
            cls[k] = c
        if len(pods) > 256:
            raise NotImplementedError(f"{len(pods)} distinct parameter sets in one intersection; the engine's table holds 256")
        return pods, cls

    def _sync_param_classes(self):
        """table and rows -> engine, when they changed (new road users, assignments to parameter attributes)"""
        pods, cls = self._param_classes()
        self._sync_param_table(pods, cls, len(self.vehicles))
        self._sync_param_rows(cls)

    def _sync_param_table(self, pods, cls, n_on_device):
        e = self._engine
        table = tuple(bytes(p) for p in pods)
        if table != self._class_table:
            if self._class_table is not None and len(table) < len(self._class_table) and n_on_device:
                e.set_agent_class(np.arange(n_on_device), cls[:n_on_device])      # rows first: none may point beyond the new table
                self._cls[:n_on_device] = cls[:n_on_device]
            e.set_param_classes(pods)
            self._class_table = table
            self._pod_bytes = table[0]

    def _sync_param_rows(self, cls):
        n = len(self.vehicles)
        ch = np.where(cls != self._cls[:n])[0]
        if ch.size:
            self._engine.set_agent_class(ch, cls[ch])
            self._cls[:n] = cls

    def add_road_user(self, user):
        """intersection.py:458-539"""
        if self.activate_sumo_cosimulation and user.follow_route:
            self._route_prototype(user)
        if self.animate:                                       # intersection.py:521-524
            if user.drawing is None:
                user.add_drawing(self.ax, **self.bicycle_drawing_kwargs)
            user.drawing.set_animated(True)
        self._attach(user)

    def get_road_user_ids(self):
        return [v.id for v in self.vehicles]

    def has_road_user(self, userId):
        assert isinstance(userId, str), "User ID has to be a string."
        return userId in self.get_road_user_ids()

    def remove_road_user(self, i_remove):
        """intersection.py:618-634"""
        self._remove([int(i_remove)])

    def remove_road_users_by_id(self, ruids):
        """intersection.py:576-616"""
        idx = [k for k, v in enumerate(self.vehicles) if v.id in ruids]
        if idx:
            self._remove(idx)

    def _remove(self, idx):
        idx = sorted(set(idx))
        self._stoch_for = -1
        for k in idx:
            if not 0 <= k < len(self.vehicles):
                raise IndexError(f"no road user {k}")
        if self._engine is not None:
            self._flush_pending()
            self._engine.remove_agents(idx)
        self._fold_all_force_logs()                  # indices are about to change
        gone = set(idx)
        keep = [k for k in range(len(self.vehicles)) if k not in gone]
        for k in idx:
            v = self.vehicles[k]
            self._detach(v)
            if v in self._pending:
                self._pending.remove(v)
            self._dirty_queues.pop(v, None)
        self.vehicles = [self.vehicles[k] for k in keep]
        m = len(keep)
        if self._S is not None:
            for a in (self._S, self._shadow, self._vd, self._ptr, self._zn, self._fx, self._fy, self._ti, self._cls):
                a[:m] = a[keep]
        if self._traj is not None and m:
            self._traj[:, :m] = self._traj[:, keep]
        for new, v in enumerate(self.vehicles):
            v._index = new
            if v._live:
                self._bind(v)
        self._drawn_stale = True
        self._scripted = [v for v in self._scripted if v._owner is self]
        self.n_bikes = len(self.vehicles)
        self._pos_stale = True

    def _bind(self, v):
        """vehicle attributes that are views of the bulk mirror"""
        k, w = v._index, type(v).N_STATES       # (a mixed population: the bulk rows are as wide as the widest class)
        v._s = self._S[k, :w]
        v.znav = self._zn[k]
        if not getattr(v, "uncontrolled", False):             # (an UncontrolledVehicle's traj is its prescription: vehicle.py:958-960)
            v.traj = self._traj[:, k, :w].T

    def _detach(self, v):
        """the vehicle leaves with private copies of everything it saw through the mirror"""
        if v._live:
            st = dict(i=v.i, destpointer=v.destpointer, force=v.force)
            v._s, v.znav, v.traj = v._s.copy(), v.znav.copy(), v.traj.copy()
            v._live = False
            v.i, v.destpointer, v.force = st["i"], st["destpointer"], st["force"]
        v._owner, v._index = None, -1

    def addEdge(self, roadEdge):
        self.road_elements.append(roadEdge)

    # ------------------------------------------------------------------ host <-> device mirror
    def _engine_ready(self):
        if not self.vehicles:
            raise RuntimeError("the intersection is empty")
        v0 = self.vehicles[0]
        if self._engine is None:
            cap = self._capacity or max(256, 4 * len(self.vehicles))
            self._engine = Engine(v0._pod(PRIORITY_RULES[self.priority_rule]), cap, device=self._device)
            self._capacity = cap
            ns = max(type(v).N_STATES for v in self.vehicles)
            self._S = np.zeros((cap, ns))
            self._shadow = np.zeros((cap, ns))
            self._vd = np.zeros(cap)
            self._ptr = np.zeros(cap, dtype=np.int32)
            self._zn = np.zeros((cap, 3), dtype=bool)
            self._fx = np.zeros(cap)
            self._fy = np.zeros(cap)
            self._ti = np.zeros(cap, dtype=np.int64)
            self._rule_on_device = self.priority_rule
            self._pod_bytes = bytes(v0._pod(PRIORITY_RULES[self.priority_rule]))
            self._class_table = (self._pod_bytes,)
            self._cls = np.zeros(cap, dtype=np.int32)
        self._flush_pending()
        return self._engine

    def _flush_pending(self):
        if not self._pending:
            return
        e = self._engine
        if len(self.vehicles) > self._capacity:
            raise RuntimeError(f"intersection capacity {self._capacity} exceeded; pass capacity= at construction")
        new = self._pending
        self._pending = []
        pods, cls = self._param_classes()                      # of everyone, the arrivals included
        ns = max(self._S.shape[1], max(type(v).N_STATES for v in new))
        if ns > self._S.shape[1]:                              # a wider vehicle class joins: wider rows
            self._widen(ns)
        self._sync_param_table(pods, cls, len(self.vehicles) - len(new))   # first: it decides the engine's state layout
        s0 = np.zeros((len(new), ns))
        for k, v in enumerate(new):
            s0[k, : v._s.size] = v._s
        vd = np.array([float(getattr(v.params, "v_desired_default", 0.0)) for v in new])   # (CarParameters have none)
        first = new[0]._index
        e.add_agents(s0[:, : e.ns], vd)
        # vehicle.traj rows of the bulk history (grown geometrically; every vehicle of an engine shares t_s)
        T = int(30 / new[0].params.t_s)                        # vehicle.py:159
        n = len(self.vehicles)
        if self._traj is None or self._traj.shape[1] < n:
            grown = np.zeros((T, max(n, 2 * (0 if self._traj is None else self._traj.shape[1])), ns))
            if self._traj is not None:
                grown[:, : self._traj.shape[1]] = self._traj
            self._traj = grown
            for v in self.vehicles:
                if v._live and not getattr(v, "uncontrolled", False):
                    v.traj = self._traj[:, v._index, : type(v).N_STATES].T
        scripted = [v for v in new if getattr(v, "uncontrolled", False)]
        if scripted:                                          # their prescribed trajectories -> the engine (csf_set_script)
            rows = [v._script if v._script is not None else np.zeros((0, 4)) for v in scripted]
            e.set_script([v._index for v in scripted], np.cumsum([0] + [r.shape[0] for r in rows]), np.vstack(rows))
            for v in scripted:                                # (what the engine holds: writes to car.traj are noticed, _push_mutations)
                v._script_sent = np.ascontiguousarray(v.traj.T).copy()
            self._scripted.extend(scripted)
        for v in new:
            if v.traj.shape[1] != T and not getattr(v, "uncontrolled", False):
                raise NotImplementedError("all road users of one intersection share t_s (one engine per intersection)")
        if scripted:
            for k, v in enumerate(new):
                if not getattr(v, "uncontrolled", False):
                    self._traj[:, first + k, : type(v).N_STATES] = v.traj.T
        elif all(type(v).N_STATES == ns for v in new):
            for c in range(0, len(new), 512):        # adopt the vehicles' own histories, a slab of rows at a time
                blk = new[c:c + 512]
                self._traj[:, first + c:first + c + len(blk), :] = np.stack([v.traj for v in blk]).transpose(2, 0, 1)
        else:
            for k, v in enumerate(new):
                self._traj[:, first + k, : type(v).N_STATES] = v.traj.T
        for k, v in enumerate(new):
            r = first + k
            self._S[r] = s0[k]
            self._shadow[r] = s0[k]
            self._vd[r] = vd[k]
            self._zn[r] = v.znav
            st = (v.i, v.destpointer, v.force)
            v._live = True
            v.i, v.destpointer, v.force = st
            self._bind(v)
            self._dirty_queues[v] = -1
        self._drawn_stale = True
        self._cls[first:first + len(new)] = 0        # (the engine starts a road user in parameter set 0)
        self._sync_param_rows(cls)

    def _widen(self, ns):
        """rows of the bulk mirror as wide as the widest vehicle class of the population"""
        for name in ("_S", "_shadow"):
            old = getattr(self, name)
            new = np.zeros((old.shape[0], ns))
            new[:, : old.shape[1]] = old
            setattr(self, name, new)
        if self._traj is not None:
            grown = np.zeros(self._traj.shape[:2] + (ns,))
            grown[:, :, : self._traj.shape[2]] = self._traj
            self._traj = grown
        for v in self.vehicles:
            if v._live:
                st = (v.i, v.destpointer, v.force)
                self._bind(v)
                v.i, v.destpointer, v.force = st

    def _mark_queue_dirty(self, v, how=None):
        """how: None rows appended, -1 queue replaced (pointer rewinds), -2 rows edited in place (pointer kept)"""
        prev = self._dirty_queues.get(v, 0)          # 0: clean
        if how == -1 or prev == -1:
            self._dirty_queues[v] = -1
        elif how == -2 or prev == -2:
            self._dirty_queues[v] = -2
        else:
            self._dirty_queues[v] = None

    def _push_mutations(self):
        """Everything the user may have changed on the Python side since the last tick."""
        e = self._engine_ready()
        n = len(self.vehicles)
        # UncontrolledVehicle: the reference reads car.traj[:, i] on every step (vehicle.py:964-979), so a trajectory written
        # after the car joined - external control - has to reach the engine: compare with what was sent, send again on a change
        changed = []
        for v in self._scripted:
            if v._live and getattr(v, "_script_sent", None) is not None:
                cur = np.ascontiguousarray(v.traj.T)
                if cur.shape != v._script_sent.shape or not np.array_equal(cur, v._script_sent):
                    v._script = cur
                    v._script_sent = cur.copy()
                    changed.append(v)
        if changed:
            e.set_script([v._index for v in changed], np.cumsum([0] + [v._script.shape[0] for v in changed]),
                         np.vstack([v._script for v in changed]))
        # destination queues (only vehicles whose setDestinations / stop / go ran since the last push):
        # replaced queues go up with reset=1, rows edited in place with reset=2, appended rows with reset=0
        if self._dirty_queues:
            groups = {1: ([], [], [0]), 2: ([], [], [0]), 0: ([], [], [0])}
            for v, how in self._dirty_queues.items():
                q = v.destqueue
                if how == -1 or v._queue_synced < 0:
                    mode, rows = 1, q
                elif how == -2:
                    mode, rows = 2, q
                elif v._queue_synced < q.shape[0]:
                    mode, rows = 0, q[v._queue_synced:]
                else:
                    continue
                agents, blocks, off = groups[mode]
                agents.append(v._index); blocks.append(rows); off.append(off[-1] + rows.shape[0])
                v._queue_synced = q.shape[0]
                v._queue_dirty = False
            self._dirty_queues = {}
            for mode in (1, 2, 0):
                agents, blocks, off = groups[mode]
                if agents:
                    e.set_dest_queue(agents, off, np.vstack(blocks), reset=mode)
        # vehicle.s edited in place (calibration.py:455-460): looked for once a vehicle.s has been handed out (Vehicle.s)
        if self._s_watched and (self._S[:n] != self._shadow[:n]).any():
            changed = np.where(np.any(self._S[:n] != self._shadow[:n], axis=1))[0]
            e.push_state(changed, self._S[changed][:, : e.ns])
            self._shadow[changed] = self._S[changed]
        if self._track_params:
            # params.v_desired_default is the per-agent parameter (demoCSFstandalone.py:104-113); the population is
            # rescanned only after some parameter object was assigned to
            if parameters.mutation_count() != self._params_seen:
                self._params_seen = parameters.mutation_count()
                vd = np.array([float(getattr(v.params, "v_desired_default", 0.0)) for v in self.vehicles])
                ch = np.where(vd != self._vd[:n])[0]
                if ch.size:
                    e.set_v_desired(ch, vd[ch])
                    self._vd[ch] = vd[ch]
                self._sync_param_classes()
        if self.priority_rule != self._rule_on_device:
            if self.priority_rule not in PRIORITY_RULES:
                raise ValueError(f"priority_rule must be one of {tuple(PRIORITY_RULES)}")
            e.set_priority_rule(PRIORITY_RULES[self.priority_rule])
            self._rule_on_device = self.priority_rule
        sig = tuple(id(el) for el in self.road_elements)
        if sig != self._road_sig:
            off, verts, F0, sg = flatten_road_elements(self.road_elements)
            e.set_road(off, verts, F0, sg)
            self._road_sig = sig
        return e

    def _pull(self, forces=True, advance=1, step=0):
        """One read-back of the device's view into the bulk mirror (vehicle.s, znav, traj are views of it); step: that many
        ticks first, in the same call (csf_step_get_tick)."""
        e = self._engine
        n = len(self.vehicles)
        if step and self._S.shape[1] == e.ns:
            # straight into the bulk mirror: the rows of the live road users are one contiguous block of it
            fx, fy = (self._fx[:n], self._fy[:n]) if forces else (None, None)
            e.step_into(step, self._S[:n], self._ptr[:n], self._zn[:n], fx, fy)
            if self._s_watched:
                self._shadow[:n] = self._S[:n]
        else:
            s, ptr, zn, fx, fy, _ = e.step_snapshot(step, forces=forces, reuse=True) if step else e.tick_snapshot(forces=forces)
            self._S[:n, : s.shape[1]] = s
            if self._s_watched:
                self._shadow[:n, : s.shape[1]] = s
            self._ptr[:n] = ptr
            self._zn[:n] = zn
            if forces:
                self._fx[:n] = fx                                     # intersection.py:860-862
                self._fy[:n] = fy
        if forces:
            self._have_force = True
        if advance:
            if forces:
                self._log_forces(fx, fy)
            T = self._traj.shape[0]
            ti = self._ti[:n] = (self._ti[:n] + advance) % T          # vehicle.py:1279-1282 (see DESIGN D5)
            if n and (ti == ti[0]).all():
                self._traj[ti[0], :n] = self._S[:n]                   # the usual case: everyone joined at tick 0
            else:
                self._traj[ti, np.arange(n)] = self._S[:n]
            if self._drawn_stale:
                self._drawn = [v for v in self.vehicles if v.drawing is not None or v.saveForces]
                self._drawn_stale = False
            for v in self._drawn:
                k = v._index
                Fx, Fy = (fx[k], fy[k]) if forces else (0.0, 0.0)
                if v.saveForces:
                    v.trajF[0, self._ti[k]] = Fx
                    v.trajF[1, self._ti[k]] = Fy
                v.update_drawing(Fres=(Fx, Fy))
        if self.activate_sumo_cosimulation:
            self.update_road_user_positions()                         # (SUMO is told every tick; else: when somebody reads them)
        else:
            self._pos_stale = True
        return fx, fy

    # vehicle.F: the per-tick magnitudes are logged as arrays and folded into a vehicle's list when it is read
    def _log_forces(self, fx, fy):
        mag = fx * fx                                                 # (np.hypot is three times the time of this at N = 16 384)
        mag += fy * fy
        self._flog.append(np.sqrt(mag, out=mag))
        if len(self._flog) >= 4096:
            self._fold_all_force_logs()

    def _fold_force_log(self, v):
        start = v._f_seen - self._flog_base
        if start < len(self._flog):
            k = v._index
            v._F.extend(float(a[k]) for a in self._flog[start:])
            v._f_seen = self._flog_base + len(self._flog)

    def _fold_all_force_logs(self):
        if self._flog:
            for v in self.vehicles:
                self._fold_force_log(v)
        self._flog_base += len(self._flog)
        self._flog = []
        for v in self.vehicles:
            v._f_seen = self._flog_base

    def _state_handed_out(self):
        """a vehicle.s view goes to the caller, who may write through it: from now on the mirror is compared with what the
        device holds (the shadow) before every tick"""
        if not self._s_watched and self._S is not None:
            self._shadow[:] = self._S                                 # (nobody has written yet: the getter runs before the write)
            self._s_watched = True

    # vehicleX / vehicleY / vehicleTheta (intersection.py:660-677) are refreshed when they are read: three column copies of the
    # mirror per tick are a tenth of the tick's host time at N = 16 384, and most ticks nobody looks
    def _positions(self):
        if self._pos_stale:
            self.update_road_user_positions()
        return self._vx, self._vy, self._vth

    vehicleX = property(lambda self: self._positions()[0], lambda self, a: setattr(self, "_vx", a))
    vehicleY = property(lambda self: self._positions()[1], lambda self, a: setattr(self, "_vy", a))
    vehicleTheta = property(lambda self: self._positions()[2], lambda self, a: setattr(self, "_vth", a))

    def update_road_user_positions(self):
        """intersection.py:660-677"""
        n = len(self.vehicles)
        self._pos_stale = False
        if self._vx.shape[0] != n:
            self._vx, self._vy, self._vth = np.zeros((n, 1)), np.zeros((n, 1)), np.zeros((n, 1))
        if n and self._S is not None and not self._pending:
            self._vx[:, 0] = self._S[:n, 0]
            self._vy[:, 0] = self._S[:n, 1]
            self._vth[:, 0] = self._S[:n, 2]
        else:
            for k, v in enumerate(self.vehicles):
                self._vx[k, 0], self._vy[k, 0], self._vth[k, 0] = v._s[0], v._s[1], v._s[2]
        if self.activate_sumo_cosimulation and n:              # intersection.py:679-688: SUMO follows the social-force positions
            move = self._traci.vehicle.moveToXY
            angles = angleSFMtoSUMO(self.vehicleTheta[:, 0])
            for k, v in enumerate(self.vehicles):
                move(v.id, "", -1, float(self.vehicleX[k, 0]), float(self.vehicleY[k, 0]), angle=float(angles[k]), keepRoute=6)

    # ------------------------------------------------------------------ reference API
    def get_untracked_foes(self):
        """intersection.py:690-745: boolean [n, n], True where a road user is not considered.  As in the reference the
        first index is the road user whose force field is evaluated (the source, whose hfov applies: :733-735) and the
        second the one that would feel it; the diagonal is True, and a single road user yields `np.array(True)`."""
        if self.n_bikes <= 1:
            return np.array((True))
        return self._push_mutations().untracked()

    def calc_forces(self):
        """intersection.py:747-864: total force on every road user from the current snapshot."""
        e = self._push_mutations()
        if self._hooked and self.n_bikes > 0:
            fx, fy = self._hooked_forces(e)
            n = len(self.vehicles)
            self._fx[:n], self._fy[:n] = fx, fy
            self._have_force = True
            self._log_forces(fx, fy)
            return fx, fy
        fx, fy = e.calc_forces()
        self._pull(forces=True, advance=0)
        self._log_forces(fx, fy)
        return fx, fy

    # ------------------------------------------------------------------ riders whose poles are drawn anew as their speed changes
    def _stochastic_riders(self):
        """the BalancingRiderBicycles with `stochastic_control_behavior=True` (parameters.py:1380-1396); cached until the population changes"""
        if self._stoch_for != len(self.vehicles) + self._stoch_epoch:
            self._stoch = [v for v in self.vehicles if getattr(v.params, "stochastic_control_behavior", False)]
            self._stoch_for = len(self.vehicles) + self._stoch_epoch
        return self._stoch

    def _resample_poles(self, riders, fx, fy):
        """BalancingRiderDynamics.step up to its gains (dynamics.py:617-647, 677-682), on the host, for the riders whose poles are
        drawn: the speed step (P controller, clamped acceleration, clamped speed - what the device is about to do with the same
        force), and where the speed changes `params.update_control_params((v_new + v) / 2)`, which draws new poles once the speed
        has moved 0.8333 m/s since the last draw.  In vehicle order: the draws come off one generator.  The tick is split for
        this - forces, draws, then the integration with the forces (csf_calc_forces / csf_apply_forces) - so that a rider steps
        with the poles the reference would have given it in THAT tick."""
        n = len(self.vehicles)
        v_now = self._S[:n, 3]
        for r in riders:
            i, p = r._index, r.params
            vd = np.sqrt(fx[i] ** 2 + fy[i] ** 2)
            a = min(max(p.k_p_v * (vd - v_now[i]), p.a_max[0]), p.a_max[1])
            v_new = min(max(v_now[i] + p.t_s * a, p.v_max_riding[0]), p.v_max_riding[1])
            if v_new != v_now[i]:
                p.update_control_params((v_new + v_now[i]) / 2)

    # ------------------------------------------------------------------ custom per-vehicle force hooks (vehicle.py:194-204, 250-299)
    def _field_engine(self, v):
        """an engine of the vehicle's class and parameter set, for its field alone (csf_pair_force: vehicle.py:1560-1648 /
        1054-1147 with THAT set's constants - the population's own engine evaluates set 0's)"""
        pod = v._pod(PRIORITY_RULES.get(self.priority_rule, 0))
        key = bytes(pod)
        eng = self._field_engines.get(key)
        if eng is None:
            from .engine import Engine

            eng = self._field_engines[key] = Engine(pod, 1)
        return eng

    def _hooked_forces(self, e):
        """calc_forces() of a population in which some vehicle carries a custom `rep_force_func` / `dest_force_func`
        (intersection.py:797-862, statement by statement): the hooks are the caller's Python, everything else comes from the
        engine's single-function entry points - csf_dest_force (queue pointer, navigation state, the class's destination force),
        csf_untracked (the mask), csf_pair_force (the class's field of one source at the receivers that track it) -, the column
        sum, the clamp and the road term are NumPy as in the reference.  O(n) device calls per tick: for the handful of road users
        scenarios with hooks have (the reference's own tick is slower still)."""
        from .utils import limitMagnitude

        n = self.n_bikes
        fdx, fdy = e.dest_force()                                  # :799 for every class force; advances the queues as the reference does
        self._pull(forces=False, advance=0)                        # vehicle.s / dest / destpointer as the hooks expect to find them
        S = self._S[:n]
        X, Y, PSI = S[:, 0].copy(), S[:, 1].copy(), S[:, 2].copy()
        for i, v in enumerate(self.vehicles):
            if v.dest_force_func is not None:                      # vehicle.py:295-297 (updateDestination has run: csf_dest_force)
                fdx[i], fdy[i] = v.dest_force_func(v)
        fx, fy = fdx.copy(), fdy.copy()
        if n > 1:
            U = e.untracked()                                      # [source, receiver], True: not considered (:690-745)
            Fx, Fy = np.zeros((n, n)), np.zeros((n, n))
            for i, v in enumerate(self.vehicles):
                trk = ~U[i]
                if not trk.any():
                    continue
                if v.rep_force_func is not None:                   # :814-820
                    fxi, fyi = v.rep_force_func(v, X[trk], Y[trk], PSI[trk])
                elif getattr(v.params, "f_0", 1.0) == 0.0 and v.MODEL != 0:
                    continue                                       # vehicle.py:1592-1593
                else:
                    fxi, fyi = self._field_engine(v).pair_force(np.r_[S[i, :3], S[i, 3]], X[trk], Y[trk], PSI[trk])
                Fx[i, trk], Fy[i, trk] = fxi, fyi                  # :822-823
            frx, fry = limitMagnitude(Fx.sum(axis=0), Fy.sum(axis=0), np.sqrt(fdx ** 2 + fdy ** 2))   # :841-845
            fx, fy = frx + fdx, fry + fdy
        for el in self.road_elements:                              # :854-857
            fxe, fye = el.calcRepulsiveForce(X[:, None], Y[:, None])
            fx = fx + np.asarray(fxe).ravel()
            fy = fy + np.asarray(fye).ravel()
        return np.ascontiguousarray(fx, dtype=float), np.ascontiguousarray(fy, dtype=float)

    def step(self):
        """intersection.py:866-896: one simulation tick of the whole population."""
        if self.animate and self.is_first_step:          # intersection.py:881-885: everyone gets a drawing on the first tick
            for v in self.vehicles:
                if v.drawing is None:
                    v.add_drawing(self.ax, animated=True, **self.bicycle_drawing_kwargs)
        self.is_first_step = False
        stochastic = self.n_bikes > 0 and self._stochastic_riders()
        if self.n_bikes > 0 and (self._hooked or stochastic):
            e = self._push_mutations()
            if self._hooked:
                fx, fy = self._hooked_forces(e)                    # intersection.py:889
            else:
                fx, fy = e.calc_forces()
            if stochastic:
                self._resample_poles(stochastic, fx, fy)
                e = self._push_mutations()                         # (a new draw is a new parameter set of its rider)
            e.apply_forces(fx, fy)                                 # :891-892: every vehicle.step(Fx[i], Fy[i])
            self._pull(forces=True, advance=1)                     # :894
        elif self.n_bikes > 0:
            self._push_mutations()
            self._pull(forces=True, advance=1, step=1)
        self.hist_n_vecs.append(self.n_bikes)

    def step_n(self, n_ticks, pull=True):
        """n_ticks ticks with no per-tick host work (the benchmark path).  `traj` receives only the final
        state; enable the engine's device-side history for dense trajectories."""
        if self.n_bikes > 0 and (self._hooked or self._stochastic_riders()):   # (custom force hooks, drawn poles: every tick passes the host)
            for _ in range(int(n_ticks)):
                self.step()
            return
        if self.n_bikes > 0 and n_ticks > 0:
            e = self._push_mutations()
            e.step(int(n_ticks))
            if pull:
                self._pull(forces=True, advance=int(n_ticks))
        self.hist_n_vecs.extend([self.n_bikes] * int(n_ticks))

    def set_animated(self, animated):
        """intersection.py:899-915: switch every drawing between blitted (animated) and ordinary artists."""
        for v in self.vehicles:
            if v.drawing is not None and hasattr(v.drawing, "set_animated"):
                v.drawing.set_animated(animated)

    @property
    def engine(self):
        return self._engine_ready()
