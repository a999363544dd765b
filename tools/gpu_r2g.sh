#!/bin/bash
TAG=${1:-r2g}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
echo "== inference bench"; timeout 300 python tools/bench_infer.py 2>&1 | grep "^{"
echo "== eager host profile"; timeout 300 python tools/probes/eager_cpu_profile.py 2>&1 | grep "eager steps"
echo "== pytest -m gpu (fast subset)"
timeout 900 python -m pytest tests -m gpu -q -x -k "not test_config1_full_size and not every_tile_pinned" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log | cut -c1-300
