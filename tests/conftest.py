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


@pytest.fixture(autouse=True)
def _cull_kernel_at_every_size(request, monkeypatch):
    """The engine picks the plain all-pairs kernel below ~3 000 road users and the cull-first kernel above
    (csf_engine.hip: pair_variant_for).  Most parity cases are small and exist to exercise the cull-first kernel - its
    classification, queue and far-field cull - so the suite pins that kernel; tests marked `auto_variant` (and every test
    that sets CSF_PAIR_VARIANT itself) run with the engine's own choice."""
    if "auto_variant" not in request.keywords and "CSF_PAIR_VARIANT" not in os.environ:
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
