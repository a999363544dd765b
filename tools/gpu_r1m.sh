#!/bin/bash
TAG=${1:-r1m}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$PWD
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
echo "== pytest -m gpu (without the full-size oracle test)"
timeout 600 python -m pytest tests -m gpu -q -x -k "not test_config1_full_size" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest_gpu.log | cut -c1-300
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
b() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 30 --warmup 6 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name: $(python -c "import json;d=json.load(open('$OUT/bench_$name.json'));print(d['ms_per_step'], d['value'], d['roofline']['achieved'])")"; tail -2 $OUT/bench_$name.err | cut -c1-200; }
b default X=1
b default2 X=1
echo "== rocprofv3 kernel trace (graph)"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 6 --no-cpu-baseline > $ROOT/$OUT/rocprof.log 2>&1 )
db=$(find $OUT/prof -name '*.db' | head -1); tr=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
src=${db:-$tr}
python tools/rocprof_summary.py $src $OUT/kernel_stats_graph.csv
rm -rf $OUT/prof
