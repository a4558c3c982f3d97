import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
