#!/bin/bash
TAG=${1:-r1o}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
echo "== ddp test (2 ranks, 1 GPU, gloo)"
timeout 600 python -m pytest tests/test_gpu_ddp.py -m gpu -q -x > $OUT/pytest_ddp.log 2>&1; echo "rc=$?"; tail -30 $OUT/pytest_ddp.log | cut -c1-300
echo "== rccl graph probe"
timeout 120 python tools/probes/rccl_graph_probe.py > $OUT/rccl_probe.log 2>&1; echo "rc=$?"; tail -5 $OUT/rccl_probe.log | cut -c1-300
