#!/bin/bash
TAG=${1:-r2a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
echo "== pinned-tile tests"
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "every_tile_pinned" > $OUT/pytest_tiles.log 2>&1; echo "rc=$?"; tail -5 $OUT/pytest_tiles.log | cut -c1-250
echo "== sweeps (h2)"
timeout 600 python tools/conv_bench.py --mode h2 --passes fwd,dgrad,wgrad --sweep --verify --iters 3 --layers conv_last,l4_conv2_d4,l4_conv3,l3_conv2_d2,l4_down > $OUT/sweep.txt 2>&1; cut -c1-150 $OUT/sweep.txt
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
b() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 30 --warmup 6 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name: $(python -c "import json;d=json.load(open('$OUT/bench_$name.json'));print(d['ms_per_step'], d['value'], d['roofline']['achieved'])")"; tail -2 $OUT/bench_$name.err | cut -c1-200; }
b a X=1
b b X=1
cp /tmp/semseg_plans_h2.json $OUT/plans_h2.json
