#!/bin/bash
# GPU box: the -m gpu suite TWICE - with the cull-first kernel pinned at every size (tests/conftest.py: most parity cases are small
# and exist to exercise its classification, queues and bounds) and on the engine's OWN choice of kernels for every population
# (CSF_TEST_AUTO_VARIANT=1: the one-wave kernel, the one-launch tick, the plain kernels, the side-by-side tick) - then (unless a run
# hung or crashed) a bench line.       tools/run_suite.sh TAG [pytest args]
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 "$@" > $OUT/pytest.log 2>&1
rc=$?
tail -5 $OUT/pytest.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: no further GPU step"; exit $rc; fi
CSF_TEST_AUTO_VARIANT=1 timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 "$@" > $OUT/pytest_auto_variant.log 2>&1
rc2=$?
tail -5 $OUT/pytest_auto_variant.log
if [ $rc2 -gt 1 ]; then echo "pytest (own choice of kernels) rc=$rc2: no further GPU step"; exit $rc2; fi
timeout -k 10 300 python bench.py --steps 1000 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err || exit 3
cat $OUT/bench.json
[ $rc -ne 0 ] && exit $rc
exit $rc2
