#!/bin/bash
# conv k-order / XCD mapping check: conv tests, bench with 1 and 2 graph copies, PMC (traffic) on the dominant conv
TAG=${1:-r1i}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$PWD
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
echo "== pytest -m gpu (without the full-size oracle test)"
timeout 600 python -m pytest tests -m gpu -q -x -k "not test_config1_full_size" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_gpu.log | cut -c1-300
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
echo "== bench h2 graph x2"
timeout 400 python bench.py --steps 30 --warmup 6 --no-cpu-baseline > $OUT/bench_g2.json 2> $OUT/bench_g2.err; echo "rc=$?"; cut -c1-200 $OUT/bench_g2.json; tail -2 $OUT/bench_g2.err
echo "== bench h2 graph x1"
SEMSEG_GRAPH_COPIES=1 timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline > $OUT/bench_g1.json 2> $OUT/bench_g1.err; cut -c1-200 $OUT/bench_g1.json
echo "== bench h2 graph x2 again"
timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline > $OUT/bench_g2b.json 2> $OUT/bench_g2b.err; cut -c1-200 $OUT/bench_g2b.json
cp /tmp/semseg_plans_h2.json $OUT/plans_h2.json
echo "== rocprofv3 kernel trace (graph x2) + gaps"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 6 --no-cpu-baseline > $ROOT/$OUT/rocprof.log 2>&1 )
db=$(find $OUT/prof -name '*.db' | head -1); tr=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
src=${db:-$tr}
python tools/rocprof_summary.py $src $OUT/kernel_stats_graph.csv
python tools/trace_gaps.py $src 0.3 | tee $OUT/trace_gaps_graph.txt
rm -rf $OUT/prof
echo "== PMC on conv_last fwd (h2, tile 5 split 4)"
MODE=h2 TILE=5 SPLIT=4 bash tools/gpu_pmc.sh $TAG/pmc conv_last fwd 2>&1 | grep -A 22 "igemm_dma"
du -sh $OUT
