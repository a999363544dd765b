#!/bin/bash
TAG=${1:-r2m}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
echo "== winograd tests"
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "winograd or weights_prepare" > $OUT/pytest_wino.log 2>&1; echo "rc=$?"; tail -12 $OUT/pytest_wino.log | cut -c1-250
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
b() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 30 --warmup 6 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name: $(python -c "import json;d=json.load(open('$OUT/bench_$name.json'));print(d['ms_per_step'], d['value'], d['config']['final_loss'])")"; tail -2 $OUT/bench_$name.err | cut -c1-200; }
b wino_1 X=1; b direct_1 SEMSEG_WINOGRAD=0; b wino_2 X=1; b direct_2 SEMSEG_WINOGRAD=0
echo "== model parity tests"
timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "not test_config1_full_size" > $OUT/pytest_models.log 2>&1; echo "rc=$?"; tail -5 $OUT/pytest_models.log | cut -c1-250
