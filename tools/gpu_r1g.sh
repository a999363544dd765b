#!/bin/bash
# One bounded gpurun call: GPU parity suite, bench A/B (side-stream wgrad on/off), graph-mode kernel trace + gap analysis,
# PMC passes on the dominant conv kernel (h2, 256x256 tile).
TAG=${1:-r1g}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$PWD
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
echo "== pytest -m gpu (without the full-size oracle test)"
timeout 600 python -m pytest tests -m gpu -q -x -k "not test_config1_full_size" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest_gpu.log | cut -c1-300
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
echo "== bench h2 (fused, side-stream wgrad)"
timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_h2.json 2> $OUT/bench_h2.err; echo "bench rc=$?"; cut -c1-330 $OUT/bench_h2.json; tail -3 $OUT/bench_h2.err
echo "== bench h2 (fused, wgrad on the main stream)"
SEMSEG_SIDE_WGRAD=0 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_h2_noside.json 2> $OUT/bench_h2_noside.err; cut -c1-330 $OUT/bench_h2_noside.json
echo "== bench h2 eager (side-stream wgrad)"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph > $OUT/bench_h2_eager.json 2> $OUT/bench_h2_eager.err; cut -c1-330 $OUT/bench_h2_eager.json
echo "== rocprofv3 kernel trace (graph replay) + gaps"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 4 --no-cpu-baseline > $ROOT/$OUT/rocprof.log 2>&1 )
echo "rocprof rc=$?"
db=$(find $OUT/prof -name '*.db' | head -1); tr=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
src=${db:-$tr}
python tools/rocprof_summary.py $src $OUT/kernel_stats_graph.csv
python tools/trace_gaps.py $src 0.3 | tee $OUT/trace_gaps_graph.txt
head -30 $OUT/kernel_stats_graph.csv | cut -c1-160
rm -rf $OUT/prof
echo "== PMC on conv_last fwd (h2, tile 5 split 4)"
MODE=h2 SEMSEG_S3_TILE=5 SEMSEG_S3_SPLITK=4 bash tools/gpu_pmc.sh $TAG/pmc conv_last fwd 2>&1 | tail -60
du -sh $OUT
