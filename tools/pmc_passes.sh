#!/bin/bash
# rocprofv3 PMC passes for one kernel of bench.py (run on the GPU box from the repo root).  Each --pmc set is its own
# run, with nothing but the counters, as MI355X_MICROARCH.md prescribes.
#   tools/pmc_passes.sh NAME [bench.py arguments ...]      -> gpurun_out/pmc_NAME/{sq1,sq2,fetch,write,grbm}
set -e
NAME=${1:-r2}; shift || true
ARGS=${@:-"--steps 20 --warmup 5"}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  name=$1; shift
  case "$ARGS" in *--preroll*) PRE="";; *) PRE="--preroll 0";; esac
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS $PRE --cpu-ticks 0 --every-pair-steps 0 > $OUT/$name.log 2>&1
  echo "pass $name done"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_BRANCH SQ_WAIT_INST_LDS
