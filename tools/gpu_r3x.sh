#!/bin/bash
TAG=${1:-r3x}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE=0
timeout 300 python -X faulthandler bench.py --config 4 --steps 5 --warmup 3 --no-cpu-baseline > $OUT/fh.json 2> $OUT/fh.err; echo rc=$?
grep -v amdgpu.ids $OUT/fh.err | head -60 | cut -c1-200
