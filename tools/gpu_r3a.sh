#!/bin/bash
# round 3, first GPU call: the whole GPU suite without xfail markers + bisect of the MobileNetV2 golden failure
TAG=${1:-r3a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
echo "== full gpu suite"
timeout 600 python -m pytest tests -m gpu -q -x --deselect "tests/test_gpu_models.py::test_native_matches_reference_golden[mnv2d_c1ds_64_train]" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.log | cut -c1-400
v() { name=$1; shift; env "$@" timeout 240 python tools/debug_golden.py mnv2d_c1ds_64_train $EXTRA > $OUT/dbg_$name.log 2>&1; echo "== $name rc=$?"; grep -v "^Loading\|amdgpu.ids" $OUT/dbg_$name.log | grep "loss\|feat\|pred\|relL2 median\|<<<\|Error\|error" | head -40 | cut -c1-200; }
EXTRA=--no-tuner v heur X=1
EXTRA=--no-tuner v heur_dw SEMSEG_DEPTHWISE_DIRECT=1
EXTRA=--no-tuner v f32 SEMSEG_CONV=f32
EXTRA=--no-tuner v nofuse SEMSEG_FUSE=0
EXTRA= v tuned_dw SEMSEG_DEPTHWISE_DIRECT=1
echo "== the failing test itself"
timeout 200 python -m pytest "tests/test_gpu_models.py::test_native_matches_reference_golden[mnv2d_c1ds_64_train]" -m gpu -q > $OUT/pytest_mnv2.log 2>&1; echo "rc=$?"; grep -n "Error\|assert\|Mismatch\|Greatest\|after-step" $OUT/pytest_mnv2.log | head -20 | cut -c1-300
