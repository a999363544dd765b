bash tools/gpu_run.sh r5q bench prof tests
bash tools/gpu_pmc_wino.sh r5q_pmc --form,6
