#!/bin/bash
TAG=${1:-r3t}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
timeout 200 python -m pytest tests/test_gpu_ddp.py -x -q -s -k "${2:-unequal}" 2>&1 | grep -a "passed\|failed\|Error\|assert\|!=" | cut -c1-400 | tee $OUT/pytest_ddp.txt
