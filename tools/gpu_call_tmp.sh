OUT=gpurun_out/r6Z; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
timeout 800 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_probes.py tests/test_gpu_eval_loop.py tests/test_gpu_drivers.py -q -x 2>&1 | grep -v amdgpu.ids | tail -8
