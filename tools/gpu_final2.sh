#!/bin/bash
# Round-end evidence inside a small GPU budget: in-box A/B of the previous library build against the current one (bench
# first, on a cool box), the full GPU parity suite + smoke, then the rocprofv3 kernel summary of the bench command.
#   gpurun --timeout 200 -- 'bash tools/gpu_final2.sh r2t'
TAG=${1:-r2t}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$PWD
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
b() { name=$1; shift; env "$@" timeout 120 python bench.py --steps 40 --warmup 6 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name: $(python -c "import json;d=json.load(open('$OUT/bench_$name.json'));print(d['ms_per_step'], d['value'])")"; }
b new1 X=1
[ -f tools/ab/libsemseg_hip_base.so ] && b base1 SEMSEG_NATIVE_LIB=$ROOT/tools/ab/libsemseg_hip_base.so
echo "== bench (full line, with cpu_baseline)"
timeout 200 python bench.py --steps 30 --warmup 6 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
cp /tmp/semseg_plans_h2.json $OUT/plans_h2.json
echo "== pytest -m gpu"
timeout 400 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log | cut -c1-300
echo "== smoke"; timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
echo "== rocprofv3 kernel trace of the bench command"
( cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 6 --no-cpu-baseline > $ROOT/$OUT/rocprof.log 2>&1 )
db=$(find $OUT/prof -name '*.db' | head -1); tr=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
src=${db:-$tr}
python tools/rocprof_summary.py $src $OUT/kernel_stats.csv
python tools/rocprof_summary.py $src $OUT/kernel_stats_by_grid.csv --by-grid
python tools/trace_gaps.py $src 0.6 > $OUT/trace_gaps.txt; head -3 $OUT/trace_gaps.txt
( cd tools && python trace_context.py $ROOT/$src copyBuffer 0.5 ) > $OUT/trace_context_copybuffer.txt 2>&1
rm -rf $OUT/prof
du -sh $OUT
