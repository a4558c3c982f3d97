#!/bin/bash
# GPU box: the -m gpu suite, then (unless the suite hung or crashed) a bench line.  tools/run_suite.sh TAG [pytest args]
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 "$@" > $OUT/pytest.log 2>&1
rc=$?
tail -15 $OUT/pytest.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: no further GPU step"; exit $rc; fi
timeout -k 10 300 python bench.py --steps 1000 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err || exit 3
cat $OUT/bench.json
exit $rc
