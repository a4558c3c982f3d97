#!/bin/bash
# GPU box: bench builds under build/ab/ with environment variants, interleaved twice.
#   tools/ab_env.sh "name1[:VAR=val,VAR=val]" "name2" ...
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2; do
  for spec in "$@"; do
    name=${spec%%:*}; envs=""
    if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
    echo -n "$spec: "
    env $envs CSF_LIB=$ROOT/build/ab/$name.so python3 $ROOT/bench.py --steps ${STEPS:-600} --warmup ${WARMUP:-30} --cpu-ticks 0 ${BENCH_ARGS} |
      grep -o '"value": [0-9.]*\|"launch_us": [0-9.]*\|"agent": [0-9.]*\|"healthy": [a-z]*' | head -4 | tr '\n' ' '
    echo
  done
done
