"""Duck-typed stand-ins for sumolib.net.Net and the traci module (neither is installed here): a four-arm junction with
one lane per arm and internal lanes, and a TraCI that records every moveToXY and serves a scripted occupancy of the
internal lanes.  Used by the host test (geometry, prototypes) and by the GPU test (the per-tick loop)."""
import numpy as np


class FakeLane:
    def __init__(self, lid, shape):
        self._id, self._shape = lid, [tuple(p) for p in shape]

    def getID(self):
        return self._id

    def getShape(self):
        return self._shape


class FakeEdge:
    def __init__(self, eid, frm, to, lanes):
        self._id, self._from, self._to, self._lanes = eid, frm, to, lanes

    def getID(self):
        return self._id

    def getLanes(self):
        return self._lanes

    def getFromNode(self):
        return self._from

    def getToNode(self):
        return self._to


class FakeNode:
    def __init__(self, nid, shape, coord):
        self._id, self._shape, self._coord = nid, shape, coord
        self.incoming, self.outgoing = [], []

    def getShape(self):
        return self._shape

    def getCoord(self):
        return self._coord

    def getIncoming(self):
        return self.incoming

    def getOutgoing(self):
        return self.outgoing


class FakeNet:
    """junction "J" at the origin, arms W / E / S / N of 60 m, 10 m footprint"""

    def __init__(self, internal=True):
        far = object()
        self.node = FakeNode("J", [(-5, -5), (5, -5), (5, 5), (-5, 5)], (0.0, 0.0))
        arms = {"W": (-1, 0), "E": (1, 0), "S": (0, -1), "N": (0, 1)}
        self.edges = []
        for name, (ax, ay) in arms.items():
            r = np.linspace(60, 5, 8)
            off = 1.6                                             # right-hand traffic: the approach lane is offset
            inn = [(ax * t - ay * off * -1, ay * t + ax * off * -1) for t in r]
            out = [(ax * t - ay * off, ay * t + ax * off) for t in r[::-1]]
            e_in = FakeEdge(name + "_in", far, self.node, [FakeLane(name + "_in_0", inn)])
            e_out = FakeEdge(name + "_out", self.node, far, [FakeLane(name + "_out_0", out)])
            self.node.incoming.append(e_in)
            self.node.outgoing.append(e_out)
            self.edges += [e_in, e_out]
        if internal:
            self.edges.append(FakeEdge(":J_0", self.node, self.node, [FakeLane(":J_0_0", [(-5, 0), (5, 0)]),
                                                                      FakeLane(":J_0_1", [(0, -5), (0, 5)])]))

    def getNode(self, nid):
        assert nid == "J"
        return self.node

    def getEdges(self):
        return self.edges


class _Vehicle:
    def __init__(self, owner):
        self.owner = owner

    def moveToXY(self, vid, edge, lane, x, y, angle=-1073741824.0, keepRoute=1):
        self.owner.moves.append((self.owner.step, vid, edge, lane, x, y, angle, keepRoute))


class _Lane:
    def __init__(self, owner):
        self.owner = owner

    def getLastStepVehicleIDs(self, lane_id):
        return tuple(self.owner.occupancy.get(lane_id, ()))


class FakeTraci:
    def __init__(self):
        self.moves, self.step, self.occupancy = [], 0, {}
        self.vehicle, self.lane = _Vehicle(self), _Lane(self)

    def simulationStep(self):
        self.step += 1
