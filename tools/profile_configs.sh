#!/bin/bash
# GPU box: rocprofv3 evidence for EVERY BASELINE config (1 - 5), each on the engine's own choice of kernels:
#   --kernel-trace --stats of tools/large_configs.py N --plain   -> <TAG>_configN_kernel_stats.csv, <TAG>_configN_kernel_trace.csv (outliers)
#   three --pmc passes (SQ counters; FETCH_SIZE; WRITE_SIZE - each its own run, nothing but the counters)
#   the config's JSON line with its roofline object (tools/large_configs.py N, HIP events inside the engine)
# Summaries land in gpurun_out/<TAG>/ ; tools/config_summary.py turns them into profiles/<TAG>_configN_roofline.json.
#   tools/profile_configs.sh TAG [configs ...]
TAG=${1:-r6_cfg}; shift
CFGS=${@:-"1 2 3 4 5"}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in $CFGS; do
  case $c in 1) T=700; P=700;; 2) T=10000; P=1000;; 3) T=1000; P=100;; 4) T=20; P=6;; 5) T=10; P=4;; esac
  echo "config $c: kernel trace ($T ticks)"
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$c -- python3 $ROOT/tools/large_configs.py $c --plain --ticks $T > $OUT/config${c}_trace.log 2>&1 || { echo "trace of config $c failed"; tail -5 $OUT/config${c}_trace.log; exit 3; }
  cp $(find $OUT/trace$c -name "*kernel_stats.csv" | head -1) $OUT/config${c}_kernel_stats.csv
  python3 $ROOT/tools/kernel_outliers.py $(find $OUT/trace$c -name "*kernel_trace.csv" | head -1) > $OUT/config${c}_kernel_outliers.txt
  rm -rf $OUT/trace$c
  for pass in "sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAIT_INST_ANY SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "fetch FETCH_SIZE" "write WRITE_SIZE"; do
    set -- $pass; name=$1; shift
    echo "config $c: pmc $name ($P ticks)"
    timeout -k 10 600 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc${c}_$name -- python3 $ROOT/tools/large_configs.py $c --plain --ticks $P > $OUT/config${c}_pmc_$name.log 2>&1 || { echo "pmc $name of config $c failed"; tail -5 $OUT/config${c}_pmc_$name.log; exit 3; }
  done
  python3 $ROOT/tools/config_summary.py $OUT $c > $OUT/config${c}_pmc.json
  rm -rf $OUT/pmc${c}_sq $OUT/pmc${c}_fetch $OUT/pmc${c}_write
  echo "config $c: bench line"
  timeout -k 10 600 python3 $ROOT/tools/large_configs.py $c > $OUT/config${c}_line.json 2> $OUT/config${c}_line.err || { echo "line of config $c failed"; tail -5 $OUT/config${c}_line.err; exit 3; }
  tail -c 400 $OUT/config${c}_line.json
done
