#!/bin/bash
# quick GPU round: parity tests (without the slow full-size oracle test unless FULL=1), bench, eager kernel profile
TAG=${1:-q}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
if [ "$FULL" = "1" ]; then K=""; else K="not test_config1_full_size"; fi
if [ "$NOTEST" != "1" ]; then timeout 900 python -m pytest tests -m gpu -q -k "$K" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log; fi
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -3 $OUT/bench.err; cp /tmp/semseg_plans.json $OUT/plans.json
if [ "$PROF" = "1" ]; then
ROOT=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 5 --warmup 3 --no-graph --no-cpu-baseline > $ROOT/$OUT/rocprof.log 2>&1 )
python tools/rocprof_summary.py $(find $OUT/prof -name '*.db' | head -1) $OUT/kernel_stats.csv; head -30 $OUT/kernel_stats.csv | cut -c1-150
rm -rf $OUT/prof
fi
