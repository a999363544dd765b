#!/bin/bash
# PMC on the dominant kernel as the bench now defines it (conv_last data gradient, tuned plan tile 8, no split-K) + bench + trace
TAG=${1:-r2q}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$PWD
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
echo "== bench"
timeout 600 python bench.py --steps 30 --warmup 6 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-200 $OUT/bench.json
cp /tmp/semseg_plans_h2.json $OUT/plans_h2.json
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 6 --no-cpu-baseline > $ROOT/$OUT/rocprof.log 2>&1 )
db=$(find $OUT/prof -name '*.db' | head -1); tr=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
src=${db:-$tr}
python tools/rocprof_summary.py $src $OUT/kernel_stats.csv
python tools/rocprof_summary.py $src $OUT/kernel_stats_by_grid.csv --by-grid
python tools/trace_gaps.py $src 0.6 > $OUT/trace_gaps.txt
rm -rf $OUT/prof
MODE=h2 TILE=8 SPLIT=1 bash tools/gpu_pmc.sh $TAG/pmc conv_last dgrad > $OUT/pmc.log 2>&1; grep -A 22 "igemm_dma" $OUT/pmc.log | head -24
