"""The C ABI under a caller that gets its arguments wrong (tests/hostile_caller.py, in a process of its own): every such call is
refused with a negative code and a message, none has a side effect, and the engine carries on."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["twod", "invpend", "planarpoint"])
def test_wrong_arguments_are_refused_without_side_effects(model):
    out = subprocess.run([sys.executable, os.path.join(HERE, "hostile_caller.py"), model], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, f"the process ended with {out.returncode} (a negative code is a signal):\n{out.stderr[-2000:]}"
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["calls"] >= 100
    assert res["accepted"] == [], f"calls that should have been refused came back >= 0: {res['accepted']}"
    assert res["silent"] == [], f"refused without a message in csf_last_error: {res['silent']}"
    assert res["same_as_twin"] and res["finite"] and res["status_flags"] == 0 and res["agents"] == 48 and res["tick"] == 16 and res["eng2_tick"] == 10
