#!/bin/bash
# round 3, final GPU call: the whole suite on the final code, bench of every BASELINE config (with cpu_baseline), inference, kernel
# statistics of the default bench command
TAG=${1:-r3z}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
echo "== bench config 1 (as the driver runs it: defaults)"
timeout 900 python bench.py > $OUT/bench_cfg1.json 2> $OUT/bench_cfg1.err; echo "rc=$?"; cut -c1-300 $OUT/bench_cfg1.json; tail -2 $OUT/bench_cfg1.err
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== full gpu suite"
timeout 2400 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log | cut -c1-300
grep -a "FAILED\|Error" $OUT/pytest_gpu.log | cut -c1-300 | head -20
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
for c in 2 3 4; do
  echo "== bench config $c"
  timeout 900 python bench.py --config $c --steps 40 --warmup 8 > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; echo "rc=$?"; cut -c1-260 $OUT/bench_cfg$c.json
done
echo "== inference bench"
timeout 300 python tools/bench_infer.py 2>&1 | grep images_per_sec | cut -c1-300 | tee $OUT/bench_infer.jsonl
echo "== kernel stats of the default bench command"
ROOT=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --no-cpu-baseline > $ROOT/$OUT/rocprof.log 2>&1 ); echo "rocprof rc=$?"
db=$(find $OUT/prof -name '*.db' | head -1); tr=$(find $OUT/prof -name '*kernel_trace.csv' | head -1); src=${db:-$tr}
python tools/rocprof_summary.py $src $OUT/kernel_stats.csv
python tools/rocprof_summary.py $src $OUT/kernel_stats_by_grid.csv --by-grid
grep -a "^{" $OUT/rocprof.log | cut -c1-300
grep "256, 256, 4, 4, 12" $OUT/kernel_stats_by_grid.csv | head -5 | cut -c1-200
rm -rf $OUT/prof
