"""Host-side angle and clamp helpers with the semantics of `cyclistsocialforce.utils` (limitAngle :124,
angleDifference :151, cart2polar :185, thresh :204, limitMagnitude :56).  Only used at construction time and
by user scripts; inside a tick the same operations run in the device code of csrc/csf_agent.hip."""
import numpy as np

TWO_PI = 2 * np.pi


def limitAngle(theta):
    """Wrap to (-pi, pi] through floor, like utils.py:124-139 (scalars and arrays)."""
    if isinstance(theta, np.ndarray):
        out = theta - TWO_PI * np.floor(theta / TWO_PI)
        out = np.where(out > np.pi, out - TWO_PI, out)
        return np.where(out < -np.pi, out + TWO_PI, out)
    out = theta - TWO_PI * np.floor(theta / TWO_PI)
    if out > np.pi:
        out -= TWO_PI
    elif out < -np.pi:
        out += TWO_PI
    return out


def angleDifference(a1, a2):
    """Signed shortest rotation from a1 to a2 (utils.py:151-182); ties resolve to the positive sense."""
    a1 = np.asarray(a1, dtype=float)
    a2 = np.asarray(a2, dtype=float)
    da = np.abs(a1 - a2)
    da = np.where(da > np.pi, TWO_PI - da, da)
    minus = np.abs(limitAngle(np.asarray(a1 - da)) - a2) < np.abs(limitAngle(np.asarray(a1 + da)) - a2)
    out = np.where(minus, -da, da)
    return float(out) if out.ndim == 0 else out


def cart2polar(x, y):
    """utils.py:185-194: rho and the arccos-based polar angle with the sign of y."""
    x = np.asarray(x, dtype=float)
    y = np.asarray(y, dtype=float)
    rho = np.hypot(x, y)
    phi = np.arccos(x / rho)
    return rho, np.where(y < 0, -phi, phi)


def thresh(x, minmax):
    """utils.py:204-227"""
    assert minmax[0] <= minmax[1], f"Minimum must be smaller then the maximum! Instead it was {list(minmax)}"
    return np.maximum(np.minimum(x, minmax[1]), minmax[0])


def limitMagnitude(x, y, r):
    """utils.py:56-86: scale vectors longer than r down to length r (in place, like the reference)."""
    rin = np.sqrt(x ** 2 + y ** 2)
    ids = rin > r
    if np.any(rin):
        x[ids] = x[ids] * r[ids] / rin[ids]
        y[ids] = y[ids] * r[ids] / rin[ids]
    return x, y


def angleSUMOtoSFM(theta):
    """utils.py:114-116"""
    return limitAngle(np.pi / 2 - np.deg2rad(theta))


def angleSFMtoSUMO(theta):
    """utils.py:119-121 (scalars and arrays): SUMO measures the heading in degrees, clockwise from north."""
    t = np.pi / 2 - np.asarray(theta, dtype=float)
    t = np.where(t < 0, t + TWO_PI, t)                 # expandAngle, utils.py:142-148
    out = np.rad2deg(t)
    return float(out) if out.ndim == 0 else out


def generateSplinePrototype(x, y, npoints=5):
    """trajectory.py:11-41: npoints samples of the interpolating cubic spline through (x, y) - the path a road user is
    given across a junction under SUMO co-simulation (host-side, once per arrival; uses scipy like the reference)."""
    from scipy import interpolate

    assert len(x) == len(y), "x and y must be same length!"
    assert len(x) >= 3, "Provide at least 3 points to calculate a cubic trajectory prototype"
    tck, _ = interpolate.splprep((x, y), s=0.0)
    x_p, y_p = interpolate.splev(np.linspace(0, 1, npoints), tck)
    return x_p, y_p
