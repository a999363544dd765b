#!/bin/bash
# HBM traffic of one whole training step by kernel (two PMC passes, separate runs as MI355X_MICROARCH.md prescribes):
#   gpurun --timeout 600 -- 'bash tools/gpu_pmc_step.sh <tag> [config] [step_ms]'
TAG=${1:-pmcstep}; CFG=${2:-1}; STEP_MS=${3:-}; ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
# launch plans: the shipped performance database (mit_semseg/perfdb: every geometry of configs[1-4] is in it, so no tuner launch is counted)
unset SEMSEG_TUNE_CACHE
cd /tmp
CMD="python $ROOT/tools/probes/step_traffic.py --config $CFG --steps 3"
run() { n=$1; shift
  timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o $n -- $CMD > $OUT/$n.log 2>&1
  echo "pmc pass $n rc=$?"; tail -1 $OUT/$n.log | cut -c1-200; }
run fetch FETCH_SIZE
grep -q "eager steps" $OUT/fetch.log || { echo "first pass did not finish"; tail -5 $OUT/fetch.log; exit 1; }
run write WRITE_SIZE
cd $ROOT
python tools/pmc_step_summary.py $OUT $STEP_MS > $OUT/summary.txt 2>&1; head -60 $OUT/summary.txt
# the floor of this decomposition (per launch: MFMA time / operand bytes / launch floor) from the same passes, before the big CSVs go
if [ "$CFG" = 1 ]; then
  python tools/abi_call_trace.py > $OUT/abi_calls.txt 2> $OUT/abi_calls.err && python tools/step_floor_model.py $OUT $OUT/abi_calls.txt > $OUT/floor_model.txt 2>&1
  head -8 $OUT/floor_model.txt
fi
find $OUT -name '*.csv' -size +2M -delete
