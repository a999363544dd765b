#!/bin/bash
# eager-mode analysis: bench A/B (side-stream wgrad on/off), eager kernel trace + gap analysis, PMC on the dominant conv
TAG=${1:-r1h}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$PWD
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
echo "== bench h2 eager, side-stream wgrad"
timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-graph > $OUT/bench_eager_side.json 2> $OUT/bench_eager_side.err; echo "rc=$?"; cut -c1-200 $OUT/bench_eager_side.json
echo "== bench h2 eager, wgrad on the main stream"
SEMSEG_SIDE_WGRAD=0 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-graph > $OUT/bench_eager_noside.json 2> $OUT/bench_eager_noside.err; cut -c1-200 $OUT/bench_eager_noside.json
echo "== rocprofv3 kernel trace (eager, side) + gaps"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-graph > $ROOT/$OUT/rocprof.log 2>&1 )
echo "rocprof rc=$?"
db=$(find $OUT/prof -name '*.db' | head -1); tr=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
src=${db:-$tr}
python tools/rocprof_summary.py $src $OUT/kernel_stats_eager.csv
python tools/trace_gaps.py $src 0.3 | tee $OUT/trace_gaps_eager.txt
rm -rf $OUT/prof
echo "== PMC on conv_last fwd (h2, tile 5 split 4)"
MODE=h2 TILE=5 SPLIT=4 bash tools/gpu_pmc.sh $TAG/pmc conv_last fwd 2>&1 | grep -A 22 "igemm_dma"
du -sh $OUT
