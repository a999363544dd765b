#!/bin/bash
# round 3, GPU call 3: segmented-executor timing probe, the re-based parity tests, driver test, config-3 steady-state bench
TAG=${1:-r3c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=/tmp/semseg_plans_h2.json
echo "== segmented probe"
timeout 600 python tools/probes/segmented_probe.py > $OUT/segmented_probe.log 2>&1; echo "rc=$?"; grep -v "amdgpu.ids\|Loading" $OUT/segmented_probe.log | tail -30 | cut -c1-250
echo "== tests"
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_drivers.py tests/test_gpu_ddp.py tests/test_gpu_depthwise.py -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log | cut -c1-300
grep -a "in units of\|updated weights\|max|dlogp|\|segmented\|2 ranks on 1 GPU\|FAILED\|Error" $OUT/pytest_gpu.log | cut -c1-300 | head -70
echo "== bench config 3 steady state"
timeout 900 python bench.py --config 3 --steps 60 --warmup 10 --no-cpu-baseline > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err; echo "rc=$?"; cut -c1-1200 $OUT/bench_cfg3.json; tail -3 $OUT/bench_cfg3.err
