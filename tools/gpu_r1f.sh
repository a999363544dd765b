#!/bin/bash
# One bounded gpurun call: new-kernel tests (all failures shown), full GPU parity suite, bench + rocprofv3 summary.
#   gpurun --timeout 1200 -- 'bash tools/gpu_r1f.sh r1f'
TAG=${1:-r1f}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$PWD
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
echo "== new-kernel tests"
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "weights_prepare or bn_h2 or conv_bn_act" > $OUT/pytest_new.log 2>&1; echo "rc=$?"; tail -40 $OUT/pytest_new.log | cut -c1-300
echo "== pytest -m gpu (without the full-size oracle test)"
timeout 600 python -m pytest tests -m gpu -q -x -k "not test_config1_full_size" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest_gpu.log | cut -c1-300
echo "== bench h2"
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_h2.json 2> $OUT/bench_h2.err; echo "bench rc=$?"; cat $OUT/bench_h2.json; tail -3 $OUT/bench_h2.err
if [ "$UNFUSED" = "1" ]; then
SEMSEG_FUSE=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_h2_unfused.json 2> $OUT/bench_h2_unfused.err; cat $OUT/bench_h2_unfused.json
fi
echo "== rocprofv3 kernel trace (h2, eager)"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 5 --warmup 3 --no-graph --no-cpu-baseline > $ROOT/$OUT/rocprof.log 2>&1 )
echo "rocprof rc=$?"
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats_h2.csv; else python tools/rocprof_summary.py $(find $OUT/prof -name '*.db' | head -1) $OUT/kernel_stats_h2.csv; fi
head -45 $OUT/kernel_stats_h2.csv | cut -c1-160
rm -rf $OUT/prof
du -sh $OUT
