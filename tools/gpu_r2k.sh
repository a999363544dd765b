#!/bin/bash
TAG=${1:-r2k}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
b() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 40 --warmup 6 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name: $(python -c "import json;d=json.load(open('$OUT/bench_$name.json'));print(d['ms_per_step'], d['value'])")"; }
b base_1 X=1; b nomp_1 SEMSEG_MULTIPOOL=0; b base_2 X=1; b nomp_2 SEMSEG_MULTIPOOL=0
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q -x -k "not test_config1_full_size and not every_tile_pinned" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log | cut -c1-300
