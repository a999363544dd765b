#!/bin/bash
# One gpurun call: GPU parity tests, bench line, rocprofv3 kernel-trace summary, per-layer conv microbench.
# Everything is written under gpurun_out/ (merged back by gpurun); summaries are copied to profiles/ by hand.
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh [tag]'
TAG=${1:-r1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
echo "== pytest -m gpu" ; timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
echo "== bench eager"; timeout 300 python bench.py --steps 10 --warmup 3 --no-graph --no-cpu-baseline > $OUT/bench_eager.json 2> $OUT/bench_eager.err; cat $OUT/bench_eager.json
echo "== conv microbench"; timeout 300 python tools/conv_bench.py --iters 5 > $OUT/conv_bench.txt 2>&1; cat $OUT/conv_bench.txt
echo "== rocprofv3 kernel trace"
ROOT=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 5 --warmup 3 --no-graph --no-cpu-baseline > $ROOT/$OUT/rocprof.log 2>&1 )
echo "rocprof rc=$?"
find $OUT/prof -name '*kernel_stats*' | head
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -40 "$f"
# keep only the small summaries (the full trace is large)
find $OUT/prof -name '*kernel_trace.csv' -size +20M -delete
du -sh $OUT
