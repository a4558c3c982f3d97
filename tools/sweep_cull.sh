#!/bin/bash
# Build-and-bench sweep of the cull kernel's tile size / occupancy target / source split (run on the GPU box).
cd $GRAFT_REPO_ROOT/cyclistsocialforce_amd/csrc
for cfg in "$@"; do
  IFS=: read tile waves split <<< "$cfg"
  rm -f csf_pair.o
  make -s FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -fno-gpu-rdc -fno-slp-vectorize -DCSF_TILE2=$tile -DCSF_CULL_WAVES=$waves" >/dev/null 2>&1
  echo -n "tile=$tile waves=$waves nsplit=$split : "
  CSF_NSPLIT=$split python3 $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --cpu-ticks 0 | grep -o '"value": [0-9.]*\|"launch_us": [0-9.]*' | tr '\n' ' '
  echo
done
