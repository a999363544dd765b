#!/bin/bash
TAG=${1:-r2h}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$PWD
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/tools/bench_configs.py --configs hrnetv2+c1 --steps 8 --warmup 5 > $ROOT/$OUT/rocprof.log 2>&1 )
db=$(find $OUT/prof -name '*.db' | head -1); tr=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
src=${db:-$tr}
python tools/rocprof_summary.py $src $OUT/kernel_stats_hrnet.csv
python tools/trace_gaps.py $src 0.4 | head -12
rm -rf $OUT/prof; grep "^{" $OUT/rocprof.log
