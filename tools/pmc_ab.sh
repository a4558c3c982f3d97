#!/bin/bash
# The SQ counter set of tools/pmc_passes.sh for several builds of the library (build/ab/NAME.so, see tools/ab.sh),
# to compare instruction counts and issue cycles of two kernel variants on one box.
set -e
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for name in "$@"; do
  OUT=$ROOT/gpurun_out/pmc_ab_$name
  mkdir -p $OUT
  export CSF_LIB=$ROOT/build/ab/$name.so
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU \
    --output-format csv -d $OUT/sq1 -- python3 $ROOT/bench.py --steps 20 --warmup 5 --cpu-ticks 0 --every-pair-steps 0 > $OUT/sq1.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_BRANCH \
    --output-format csv -d $OUT/sq2 -- python3 $ROOT/bench.py --steps 20 --warmup 5 --cpu-ticks 0 --every-pair-steps 0 > $OUT/sq2.log 2>&1
  echo "== $name"
  python3 $ROOT/tools/pmc_summary.py $OUT
done
