#!/usr/bin/env python3
"""Where does a kernel touch scratch memory?  Compiles csf_pair.hip (or the file given) to gfx950 assembly and lists, for
one kernel, every scratch_load / scratch_store with the loops (backward branches) that contain it.

    tools/isa_spills.py [mangled-kernel-name-substring] [source.hip]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needle = sys.argv[1] if len(sys.argv) > 1 else "pair_cull_kernelILb0ELb1ELb0ELb1ELi32ELb1ELi8E"
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "cyclistsocialforce_amd", "csrc", "csf_pair.hip")
out = "/tmp/isa_spills.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
                       "-fno-slp-vectorize", "-w", "-S", "--cuda-device-only", "-o", out, src])
s = open(out).read()
m = re.search(r"^(_Z\w*%s\w*):" % re.escape(needle), s, re.M)
if not m:
    sys.exit("no kernel matches " + needle)
a = m.start()
body = s[a:s.index("s_endpgm", a) + 10]
labels, ins = {}, []
for l in body.split("\n"):
    l = l.strip()
    lm = re.match(r"^(\.LBB\d+_\d+):", l)
    if lm:
        labels[lm.group(1)] = len(ins)
        continue
    if not l or l.startswith((".", ";")) or l.endswith(":"):
        continue
    ins.append(l)
loops = []
for i, l in enumerate(ins):
    bm = re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if bm and bm.group(1) in labels and labels[bm.group(1)] <= i:
        loops.append((labels[bm.group(1)], i))
print(m.group(1))
print(f"{len(ins)} instructions, {sum('scratch_' in l for l in ins)} scratch, {sum(l.startswith('v_') for l in ins)} VALU, "
      f"{sum(l.startswith('s_') for l in ins)} SALU, {sum(l.startswith('ds_') for l in ins)} LDS, {sum(l.startswith('global_') for l in ins)} global")
print("loops (first, last instruction):", loops)
for i, l in enumerate(ins):
    if "scratch_" in l:
        inner = [lp for lp in loops if lp[0] <= i <= lp[1]]
        depth = len(inner)
        smallest = min(inner, key=lambda lp: lp[1] - lp[0]) if inner else None
        print(f"  {i:5d} {l.split()[0]:24s} loop depth {depth} innermost {smallest}")
