"""cyclistsocialforce_amd/csrc/csf_math64.h on the CPU: the header the per-agent kernel takes its fp64 sqrt / division /
sincos / tan / atan2 from is compiled with g++ (hardware approximations replaced by libm results cut to 25 bits) and
compared with libm over the arguments the kernel has (tests/native/math64_harness.cpp)."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_math64_against_libm(tmp_path):
    exe = str(tmp_path / "m64")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "cyclistsocialforce_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "math64_harness.cpp"), "-o", exe], check=True)
    r = json.loads(subprocess.run([exe], check=True, capture_output=True, text=True).stdout)
    assert r["zeros"]
    # lengths and quotients: to the last bit or two
    assert r["ulp_sqrt"] <= 1 and r["ulp_div"] <= 1 and r["ulp_rcp"] <= 2 and r["ulp_rsqrt"] <= 2
    # sin / cos: absolute (next to a zero the two-constant reduction keeps 1e-16 absolute, which is what x += v t cos(psi) sees)
    assert r["abs_sin"] < 2.5e-16 and r["abs_cos"] < 2.5e-16
    assert r["ulp_tan"] <= 4 and r["ulp_atan2"] <= 2 and r["abs_atan2"] < 9e-16
