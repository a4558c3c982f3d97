#!/bin/bash
# GPU box: the round's evidence for bench.py at its headline configuration, all from ONE build of the library:
#   PMC passes of the pair kernel (tools/pmc_passes.sh), rocprofv3 --kernel-trace --stats of the driver's command,
#   the bench line with the driver's command and with the default 1000 steps.   tools/profile_round.sh TAG
set -e
TAG=${1:-r3_v1}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
bash $ROOT/tools/pmc_passes.sh $TAG --steps 200 --warmup 20 > $OUT/pmc_passes.log 2>&1
python3 $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/pmc_$TAG pair_cull_kernel "N=16384 TwoDBicycle, bench.py --steps 200 --warmup 20 (warm: ~0.1 s pre-roll)" > $OUT/pair_kernel_pmc.json
# bench.py takes roofline.traffic from the newest profiles/r*_v*_pair_kernel_pmc.json of the SAME build: put it there first
case $TAG in r[0-9]*_v[0-9]*) cp $OUT/pair_kernel_pmc.json $ROOT/profiles/${TAG}_pair_kernel_pmc.json ;; esac
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 20 --warmup 5 --cpu-ticks 0 --every-pair-steps 0 > $OUT/trace.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cd $ROOT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
