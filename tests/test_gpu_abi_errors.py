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


_BOUNDARY = r"""
import ctypes as C, json, sys
sys.path.insert(0, %r)
from cyclistsocialforce_amd import _ffi, parameters
L = _ffi.load()
pod = parameters.default_pod("twod")
h = L.csf_create_v(C.byref(pod), C.sizeof(pod), _ffi.ABI_VERSION, 64, 0)
rc = L.csf_sync(h)
msg = L.csf_last_error(h).decode()
print(json.dumps({"rc": rc, "msg": msg, "destroy": L.csf_destroy(h)}))
"""


@pytest.mark.gpu
@pytest.mark.parametrize("kind, word", [("1", "bad_alloc"), ("2", "CSF_DEBUG_THROW=2")])
def test_no_exception_crosses_the_c_boundary(kind, word):
    """every entry point is a function-try-block (csrc/engine/consts.inc: csf_caught): CSF_DEBUG_THROW makes csf_sync throw, the caller gets
    CSF_E_HOST = -7 and a sentence, and the engine can still be destroyed"""
    env = dict(os.environ, CSF_DEBUG_THROW=kind)
    out = subprocess.run([sys.executable, "-c", _BOUNDARY % os.path.dirname(HERE)], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["rc"] == -7 and word in res["msg"] and res["destroy"] == 0, res
