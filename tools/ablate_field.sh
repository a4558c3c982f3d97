#!/bin/bash
# Timing-only ablation (run on the GPU box): the pair kernel without the field arithmetic, i.e. tile fill +
# classification + tests + queue traffic + reductions.  Outputs are wrong by construction; only the time counts.
cd $GRAFT_REPO_ROOT/cyclistsocialforce_amd/csrc
rm -f csf_pair.o
make -s FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -fno-gpu-rdc -fno-slp-vectorize -DCSF_SKIP_FIELD" >/dev/null 2>&1
echo -n "without field arithmetic: "
python3 $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --cpu-ticks 0 | grep -o '"launch_us": [0-9.]*'
rm -f csf_pair.o
make -s >/dev/null 2>&1
echo -n "full kernel:              "
python3 $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --cpu-ticks 0 | grep -o '"launch_us": [0-9.]*'
