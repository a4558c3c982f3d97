#!/bin/bash
# Timing-only ablation (GPU box): pair kernel with only start-up + tile fill + reduction (no classification, no tests,
# no field).  Outputs are wrong by construction.
cd $GRAFT_REPO_ROOT/cyclistsocialforce_amd/csrc
rm -f csf_pair.o
make -s FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -fno-gpu-rdc -fno-slp-vectorize -DCSF_SKIP_LOOP" >/dev/null 2>&1
echo -n "start-up + fill + reduce only: "
python3 $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --cpu-ticks 0 | grep -o '"launch_us": [0-9.]*'
rm -f csf_pair.o
make -s >/dev/null 2>&1
