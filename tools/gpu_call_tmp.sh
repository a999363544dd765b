export TMPDIR=/tmp
bash tools/gpu_run.sh r6M bench "test:env_switch_keeps_model_parity and FUSE" 2>&1 | tail -12
bash tools/gpu_pmc_wino.sh r6M_pmc_fused --form,8 2>&1 | tail -40
bash tools/gpu_pmc_wino.sh r6M_pmc_l4fwd --mode,fwd,--geom,2:64:64:512:512:4,--tile,14 2>&1 | tail -60
bash tools/gpu_run.sh r6M_prof prof prof:4 2>&1 | grep -v "^wrote" | tail -80
