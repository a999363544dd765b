OUT=gpurun_out/r6e; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_run.sh r6e bench 2>&1 | tail -8
bash tools/gpu_ab_trees.sh r6e_ab build_ab/r4ab 2>&1 | tail -12
