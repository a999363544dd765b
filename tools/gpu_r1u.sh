#!/bin/bash
TAG=${1:-r1u}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
echo "== pytest -m gpu (all but the full-size oracle test)"
timeout 900 python -m pytest tests -m gpu -q -x -k "not test_config1_full_size" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_gpu.log | cut -c1-300
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
b() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 30 --warmup 6 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name: $(python -c "import json;d=json.load(open('$OUT/bench_$name.json'));print(d['ms_per_step'], d['value'], d['roofline']['achieved'])")"; tail -2 $OUT/bench_$name.err | cut -c1-200; }
b a X=1
b b X=1
cp /tmp/semseg_plans_h2.json $OUT/plans_h2.json
