import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "auto_variant: the engine chooses its pair kernel by population size")
    config.addinivalue_line("markers", "cull_variant: written around the cull-first kernel (its counters, its name): pinned even with CSF_TEST_AUTO_VARIANT=1")


def pytest_generate_tests(metafunc):
    """Every GPU test against the reference's golden vectors runs TWICE in one pass of the suite: with the cull-first kernel pinned
    (below) and on the engine's own choice of kernels for its population - the one-wave kernel, the one-launch tick, the plain
    kernels -, which is what a drop-in user of those population sizes gets.  (Tests that are marked `auto_variant` or `cull_variant`,
    or that pin a kernel themselves, keep their single run.)"""
    fn = metafunc.function
    marks = {m.name for m in metafunc.definition.iter_markers()}
    if "golden" in fn.__name__ and "gpu" in marks and not ({"auto_variant", "cull_variant"} & marks) and "kernel_choice" not in metafunc.fixturenames:
        metafunc.fixturenames.append("kernel_choice")
        metafunc.parametrize("kernel_choice", ["cull-first", "own-choice"])


@pytest.fixture
def kernel_choice(request):
    return getattr(request, "param", "cull-first")


@pytest.fixture(autouse=True)
def _cull_kernel_at_every_size(request, monkeypatch):
    """The engine picks the plain all-pairs kernel below ~3 000 road users and the cull-first kernel above
    (csf_engine.hip: pair_variant_for).  Most parity cases are small and exist to exercise the cull-first kernel - its
    classification, queue and far-field cull - so the suite pins that kernel; tests marked `auto_variant` (and every test
    that sets CSF_PAIR_VARIANT itself) run with the engine's own choice."""
    if "auto_variant" in request.keywords or "CSF_PAIR_VARIANT" in os.environ:
        return
    cs = getattr(request.node, "callspec", None)
    if cs is not None and cs.params.get("kernel_choice") == "own-choice":
        return
    # CSF_TEST_AUTO_VARIANT=1: the whole suite on the engine's own choice, but for the tests written around the cull-first kernel
    if os.environ.get("CSF_TEST_AUTO_VARIANT") != "1" or "cull_variant" in request.keywords:
        monkeypatch.setenv("CSF_PAIR_VARIANT", "0")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))

    return load


def hetero_classes(g, model):
    """The parameter sets of tests/golden/hetero.npz (make_golden.py: HETERO_RECIPES) as csf_params PODs, built by the
    host mirror of the reference's parameter classes from the same constructor keywords, + the set of every vehicle."""
    import json

    from cyclistsocialforce_amd import parameters

    recipes = json.loads(str(g[f"{model}_recipes"]))
    return [parameters.default_pod(model, **kw) for kw in recipes], g[f"{model}_cls"].astype("uint8")


MIXED_OWN = {"twod": dict(hfov=1.0, f_0=10.0), "bicycle": dict(hfov=1.1 * 3.141592653589793, p_0=40.0, p_decay=4.0),
             "invpend": dict(hfov=2.5, e_0=0.9, k_p_v=12.0), "planarpoint": dict(hfov=1.5, f_0=5.0, poles=[-3.0 + 0j]),
             "planarbike": dict(hfov=2.8, sigma_0=0.6)}


def mixed_classes(g):
    """The parameter sets of tests/golden/mixed.npz (make_golden.py: gen_mixed): the defaults of each of the five vehicle
    classes + the sets the last five vehicles own, as csf_params PODs, and the set of every vehicle."""
    from cyclistsocialforce_amd import parameters

    order = ["twod", "bicycle", "invpend", "planarpoint", "planarbike"]
    pods = [parameters.default_pod(m) for m in order] + [parameters.default_pod(m, **MIXED_OWN[m]) for m in order]
    cls = [order.index(str(m)) + (5 if own else 0) for m, own in zip(g["models"], g["own"])]
    import numpy as np
    return pods, np.array(cls, dtype="uint8")


def shadow_run(e, pop, ticks, window, traj_len=3000):
    """The engine runs `ticks` ticks without interruption while the oracle SHADOWS it in windows of `window` ticks.

    The dynamics of a crowd is chaotic (two fp64 runs started 4e-6 m apart are decimetres apart after 1 000
    ticks), and every discontinuity of the force law (the field-of-view edge, sign(phi), the navigation state machine, the
    walk / ride switch) turns a rounding difference of the STATE into a visible one, so a free run of engine and oracle side
    by side compares the two programs only until the first such event.  Here the oracle is re-anchored on the engine's
    state of two consecutive ticks at every window boundary (two: the planner reads the previous position from the
    trajectory ring, tests/test_gpu_large.py::test_config2), and every window compares `window` ticks of both programs
    from a common state.  The common state includes what vehicle.s does not carry - vehicle.x of an InvPendulumBicycle
    (steer and lean RATES, the unwrapped yaw) and vehicle.zrid: without them the oracle keeps its own rates behind the
    engine's angles, and the stiff roll loop of a slow rider (gains ~ v^-3, parameters.py:1832-1892) turns that mismatch
    into decimetres within one window (tools/invpend_cut_study.py shows it with the oracle alone).
    Returns (largest position deviation at a window end, per-road-user deviation at the last one, the engine's and the
    oracle's final state)."""
    import numpy as np

    def anchor():
        s, ptr, zn, t = e.state(with_nav=True)
        pop.push_state(s, ptr, zn, col=t % traj_len)
        x, _, zrid = e.integrator_state()                      # include/csf.h: csf_get_integrator_state
        pop.set_lti(x, zrid)
        return t

    tick, worst, dev, ref = 0, 0.0, None, None
    while tick < ticks:
        end = min(tick + window, ticks)
        if tick > 0:                                          # anchor on the engine's states of ticks `tick - 1` (where it is) and `tick`
            assert anchor() == tick - 1
            e.step(1); pop.step(1)
            anchor()
        last = end == ticks
        k = end - tick - (0 if last else 1)                   # stop one tick short of the next boundary
        e.step(k); pop.step(k)
        got, ref = e.state(), pop.state()
        dev = np.hypot(got[:, 0] - ref[:, 0], got[:, 1] - ref[:, 1])
        worst = max(worst, float(dev.max()))
        tick = end
    return worst, dev, got, ref
