#!/bin/bash
# First GPU call of the next round (≈ 4-5 min): everything that was written after the previous round's GPU budget was spent.
#   1. the pending tests (tests/test_gpu_zz_*.py: input pipeline, depthwise kernels, new backbones, evaluation loop) -- run
#      with the xfail marker disabled so that failures show as failures
#   2. in-box A/B of the queued one-line experiments
#   3. depthwise kernels on MobileNetV2 (SEMSEG_DEPTHWISE_DIRECT=1) through the golden parity case
#   gpurun --timeout 420 -- 'bash tools/gpu_next_round_first.sh r3a'
TAG=${1:-r3a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
echo "== pending tests (markers ignored: --runxfail)"
timeout 300 python -m pytest tests/test_gpu_zz_depthwise.py tests/test_gpu_zz_input.py tests/test_gpu_zz_models.py -m gpu -q --runxfail \
    > $OUT/pytest_pending.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_pending.log | cut -c1-300
echo "== MobileNetV2 golden on the depthwise kernels"
SEMSEG_DEPTHWISE_DIRECT=1 timeout 200 python -m pytest "tests/test_gpu_zz_models.py::test_new_backbones_match_reference_golden[mnv2d_c1ds_64_train]" \
    -m gpu -q --runxfail > $OUT/pytest_mnv2_direct.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_mnv2_direct.log | cut -c1-300
echo "== ResNeXt golden on the grouped kernels"
SEMSEG_GROUPED_DIRECT=1 timeout 200 python -m pytest "tests/test_gpu_zz_models.py::test_new_backbones_match_reference_golden[resnext101_upernet_128_eval]" \
    -m gpu -q --runxfail > $OUT/pytest_resnext_direct.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_resnext_direct.log | cut -c1-300
echo "== queued A/B (bench, interleaved, 2 rounds)"
bash tools/gpu_ab.sh $TAG/ab base:X=1 wino512:SEMSEG_WINOGRAD_MIN_C=512 \
    wsplit:SEMSEG_WGRAD_MAX_SPLIT=256,SEMSEG_TUNE_CACHE=/tmp/plans_wsplit.json
