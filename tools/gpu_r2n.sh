#!/bin/bash
TAG=${1:-r2n}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$PWD
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 6 --no-cpu-baseline > $ROOT/$OUT/rocprof.log 2>&1 )
db=$(find $OUT/prof -name '*.db' | head -1); tr=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
src=${db:-$tr}
python tools/rocprof_summary.py $src $OUT/kernel_stats_by_grid.csv --by-grid
python tools/rocprof_summary.py $src $OUT/kernel_stats.csv
rm -rf $OUT/prof
grep -i "wino" $OUT/kernel_stats_by_grid.csv | cut -c1-170
