#!/usr/bin/env python3
"""The one-launch tick under rocprofv3 --kernel-trace --stats (run this script after `--`): N road users, 3 000 ticks.
    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/mid_kernel_duration.py 1024 200"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synthetic_population  # noqa: E402
from cyclistsocialforce_amd import parameters  # noqa: E402
from cyclistsocialforce_amd.engine import Engine  # noqa: E402

n, box = int(sys.argv[1]), float(sys.argv[2])
s0, off, dq = synthetic_population(n, box, reach=tuple(50.0 * k for k in range(1, 14)))
e = Engine(parameters.default_pod("twod"), n)
e.add_agents(s0, 5.0)
e.set_dest_queue(np.arange(n), off, dq, reset=True)
e.step(3000, sync=True)
print("one-launch ticks", e.mid_ticks())
e.close()
