#!/bin/bash
# Kernel A/B on ONE box: tools/ab.sh build NAME   (here: current tree -> build/ab/NAME.so)
#                        tools/ab.sh run NAME...  (GPU box: bench every build, twice, interleaved)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  mkdir -p $ROOT/build/ab
  make -s -C $ROOT/cyclistsocialforce_amd/csrc
  cp $ROOT/cyclistsocialforce_amd/libcsf_hip.so $ROOT/build/ab/$2.so
  exit 0
fi
shift
for rep in 1 2; do
  for name in "$@"; do
    echo -n "$name: "
    CSF_LIB=$ROOT/build/ab/$name.so python3 $ROOT/bench.py --steps ${STEPS:-600} --warmup ${WARMUP:-30} --cpu-ticks 0 ${BENCH_ARGS} |
      grep -o '"value": [0-9.]*\|"launch_us": [0-9.]*\|"healthy": [a-z]*' | tr '\n' ' '
    echo
  done
done
