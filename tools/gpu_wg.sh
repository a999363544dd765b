#!/bin/bash
# Winograd weight gradient: the new parity tests, then an in-box A/B of the bench (base, wgrad, base)
TAG=${1:-wg}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
timeout 200 python -m pytest tests/test_gpu_ops.py -q -x -k "winograd" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
b() { name=$1; shift; env "$@" timeout 200 python bench.py --steps 40 --warmup 6 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name: $(python -c "import json;d=json.load(open('$OUT/bench_$name.json'));print(d['ms_per_step'], d['value'])")"; }
b base1 X=1
b wg1 SEMSEG_WINOGRAD_WGRAD=1
b base2 X=1
b wg2 SEMSEG_WINOGRAD_WGRAD=1
