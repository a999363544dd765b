#!/bin/bash
# round 3: kernel statistics + trace gaps of the HRNetV2 step with its branches on side streams (bench --config 4)
TAG=${1:-r3ae}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$PWD
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/plans_c4.json
timeout 300 python bench.py --config 4 --steps 10 --warmup 6 --no-cpu-baseline > $OUT/bench_c4.json 2>/dev/null; cut -c1-160 $OUT/bench_c4.json
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --config 4 --steps 10 --warmup 6 --no-cpu-baseline > $ROOT/$OUT/rocprof.log 2>&1 ); echo "rocprof rc=$?"
db=$(find $OUT/prof -name '*.db' | head -1); tr=$(find $OUT/prof -name '*kernel_trace.csv' | head -1); src=${db:-$tr}
python tools/rocprof_summary.py $src $OUT/kernel_stats.csv
python tools/trace_gaps.py $src 0.3 > $OUT/trace_gaps.txt; head -10 $OUT/trace_gaps.txt
rm -rf $OUT/prof
