#!/bin/bash
# One bounded gpurun call (≈15 min): accuracy check of the conv schemes, GPU parity tests, bench (h2 default) + rocprofv3 summary.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r1e.sh r1e'
TAG=${1:-r1e}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$PWD
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
echo "== accuracy (h2 / s3 / f32 / torch cpu32 vs float64)"
timeout 240 python tools/s3_check.py > $OUT/s3_check.txt 2>&1; echo "s3_check rc=$?"; tail -30 $OUT/s3_check.txt
echo "== pytest -m gpu (without the full-size oracle test)"
timeout 600 python -m pytest tests -m gpu -q -x -k "not test_config1_full_size" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
echo "== bench h2"
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_h2.json 2> $OUT/bench_h2.err; echo "bench rc=$?"; cat $OUT/bench_h2.json; tail -3 $OUT/bench_h2.err
cp /tmp/semseg_plans_h2.json $OUT/plans_h2.json 2>/dev/null
echo "== rocprofv3 kernel trace (h2, eager)"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 5 --warmup 3 --no-graph --no-cpu-baseline > $ROOT/$OUT/rocprof.log 2>&1 )
echo "rocprof rc=$?"
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats_h2.csv; head -40 "$f" | cut -c1-200; else python tools/rocprof_summary.py $(find $OUT/prof -name '*.db' | head -1) $OUT/kernel_stats_h2.csv; head -40 $OUT/kernel_stats_h2.csv | cut -c1-200; fi
rm -rf $OUT/prof
du -sh $OUT
