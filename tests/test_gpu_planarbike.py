"""PlanarBicycle (vehicle.py:2031-2076) on the HIP path: against the fixtures captured from the
reference and against the CPU oracle.  Needs a real MI355X; everything goes through the C ABI."""
import numpy as np
import pytest

from oracle import csf_oracle as orc

pytestmark = pytest.mark.gpu


def engine(s0, vdes, off, dq, **over):
    from cyclistsocialforce_amd import parameters
    from cyclistsocialforce_amd.engine import Engine

    s0 = np.asarray(s0, dtype=float)
    e = Engine(parameters.default_pod("planarbike", **over), s0.shape[0])
    e.add_agents(s0, vdes)
    e.set_dest_queue(np.arange(s0.shape[0]), off, dq, reset=True)
    return e


def test_single_steps_golden(golden):
    """PlanarTwoWheelerDynamics.step (dynamics.py:225-258): 120 random (state, force) pairs, two consecutive steps.  The
    kernel's speed-independent closed form (csf_engine.hip: derive_planarbike) against the reference, which places the
    poles and simulates a 10-s step response for every single step."""
    g = golden("planarbike")
    s01, F = g["steps_s01"], g["steps_F01"]
    n = s01.shape[0]
    e = engine(s01[:, :5], 5.0, np.arange(n + 1), np.c_[s01[:, 0], s01[:, 1], np.zeros(n)])
    e.apply_forces(F[:, 0], F[:, 1])
    np.testing.assert_allclose(e.state(), s01[:, 5:], rtol=1e-11, atol=1e-12)
    e.apply_forces(F[:, 2], F[:, 3])
    np.testing.assert_allclose(e.state(), g["steps_s2"], rtol=1e-11, atol=1e-12)
    assert (e.status() == 0).all()


@pytest.mark.parametrize("tag", ["demo", "dense"])
def test_population_trajectories_golden(golden, tag):
    """intersection.py:866-896 end to end: the demo geometry for 700 ticks and a dense population for 150."""
    g = golden("planarbike")
    e = engine(g[f"{tag}_s0"], g[f"{tag}_vdes"], g[f"{tag}_off"], g[f"{tag}_dq"])
    S = g[f"{tag}_S"]
    extent = max(np.ptp(S[..., 0]), np.ptp(S[..., 1]), 1.0)
    worst = 0.0
    for k in range(1, S.shape[0]):
        e.step(10)
        got = e.state()
        worst = max(worst, np.abs(got[:, :2] - S[k][:, :2]).max() / extent)
        np.testing.assert_allclose(got[:, :2], S[k][:, :2], rtol=0, atol=1e-4 * extent, err_msg=f"{tag} sample {k}")
        np.testing.assert_allclose(got[:, 3:], S[k][:, 3:], rtol=0, atol=2e-3, err_msg=f"{tag} sample {k}")
    assert (e.status() == 0).all()
    print(f"planarbike {tag}: worst position deviation / extent = {worst:.2e}")


def test_random_population_vs_oracle():
    import bench

    n, box, ticks = 768, 140.0, 100
    s0, off, dq = bench.synthetic_population(n, box, seed=6)
    e = engine(s0, 5.0, off, dq)
    pop = orc.Population(orc.default_params("planarbike"), s0, 5.0, off, dq)
    e.step(ticks)
    pop.step(ticks)
    got, ref = e.state(), pop.state()
    dev = np.abs(got[:, :2] - ref[:, :2]).max(axis=1)
    print(f"planarbike N={n}: max |dpos| / box after {ticks} ticks = {dev.max() / box:.2e}")
    assert dev.max() < 1e-4 * box
    assert (e.status() == 0).all()


def test_mirror_class_and_uncontrollable_flag():
    from cyclistsocialforce_amd import _ffi
    from cyclistsocialforce_amd.intersection import SocialForceIntersection
    from cyclistsocialforce_amd.parameters import PlanarBicycleParameters
    from cyclistsocialforce_amd.vehicle import PlanarBicycle

    a = PlanarBicycle((0, 0, 0, 4, 0), id="a")
    b = PlanarBicycle((0, 6, 0, 4, 0.02, 9, 9), id="b")             # longer initial states are truncated (vehicle.py:149-152)
    assert isinstance(a.params, PlanarBicycleParameters) and b.s.shape == (5,)
    a.setDestinations((40.0, 80.0, 81.0), (0.0, 0.0, 0.0))
    b.setDestinations((40.0, 80.0, 81.0), (6.0, 6.0, 6.0))
    ins = SocialForceIntersection((a, b))
    for _ in range(200):
        ins.step()
    assert a.s[0] > 7.0 and abs(a.s[1]) < 1.0 and a.i == 200 and np.isfinite(b.s).all()
    with pytest.raises(AssertionError):
        PlanarBicycle((0, 0, 0, 4))
    # standing still the reference's pole placement fails ("System not controllable!", dynamics.py:1212-1214): flagged
    e = engine(np.array([[0.0, 0, 0, 0.0, 0]]), 5.0, [0, 1], [[10.0, 0.0, 0.0]])
    e.apply_forces([1.0], [0.5])
    assert e.status()[0] & _ffi.ST_UNCONTROLLABLE and np.isfinite(e.state()).all()
