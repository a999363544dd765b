#!/bin/bash
# Follow-up to gpu_final2.sh inside ~80 s: library A/B (bench first), the GPU parity suite without the tile-pinned sweep
# (unchanged kernels, 174 cases), full bench line, rocprofv3 kernel summary of the bench command.
TAG=${1:-r2u}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$PWD
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
b() { name=$1; shift; env "$@" timeout 100 python bench.py --steps 40 --warmup 6 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name: $(python -c "import json;d=json.load(open('$OUT/bench_$name.json'));print(d['ms_per_step'], d['value'])")"; }
b new1 X=1
[ -f tools/ab/libsemseg_hip_base.so ] && b base1 SEMSEG_NATIVE_LIB=$ROOT/tools/ab/libsemseg_hip_base.so
echo "== bench (full line, with cpu_baseline)"
timeout 100 python bench.py --steps 30 --warmup 6 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
cp /tmp/semseg_plans_h2.json $OUT/plans_h2.json
echo "== pytest -m gpu (without the tile-pinned sweep)"
timeout 200 python -m pytest tests -m gpu -q -x -k "not every_tile_pinned" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log | cut -c1-300
echo "== rocprofv3 kernel trace of the bench command"
( cd /tmp && timeout 100 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 6 --no-cpu-baseline > $ROOT/$OUT/rocprof.log 2>&1 )
db=$(find $OUT/prof -name '*.db' | head -1); tr=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
src=${db:-$tr}
python tools/rocprof_summary.py $src $OUT/kernel_stats.csv
python tools/rocprof_summary.py $src $OUT/kernel_stats_by_grid.csv --by-grid
python tools/trace_gaps.py $src 0.6 > $OUT/trace_gaps.txt; head -3 $OUT/trace_gaps.txt
rm -rf $OUT/prof
