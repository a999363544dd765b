#!/bin/bash
TAG=${1:-r1z}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
echo "== pinned-tile tests"
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "every_tile_pinned" > $OUT/pytest_tiles.log 2>&1; echo "rc=$?"; tail -5 $OUT/pytest_tiles.log | cut -c1-250
echo "== fwd/dgrad sweep (h2)"
timeout 600 python tools/conv_bench.py --mode h2 --passes fwd,dgrad --sweep --verify --iters 3 --layers conv_last,deepsup,l4_conv2_d4,l4_conv3,l3_conv2_d2,l4_conv1,l4_down,l3_conv3 > $OUT/fwd_sweep.txt 2>&1; cut -c1-1300 $OUT/fwd_sweep.txt
