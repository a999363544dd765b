bash tools/gpu_run.sh r5j "test:knife_edge or all_taps or (full_size and 456x680)"
timeout 900 python tools/conv_bench.py --mode h2 --passes fwd,dgrad --sweep --verify --layers l3_conv2_d2,l3_conv3,l3_conv1,l4_conv3,l4_conv1,l2_conv2,l1_conv2,stem_conv2,hr_48,hr_96,hr_192,hr_384 > gpurun_out/r5j/fwd_dgrad_sweep.txt 2>&1
timeout 600 python tools/conv_bench.py --mode h2 --passes wgrad --sweep --verify --layers l3_conv2_d2,l3_conv3,l3_conv1,l4_conv3,l4_conv1,l2_conv2,hr_192,hr_384,deepsup,l4_conv2_d4 > gpurun_out/r5j/wgrad_sweep.txt 2>&1
grep -v amdgpu gpurun_out/r5j/fwd_dgrad_sweep.txt | cut -c1-120 | tail -30
grep -v amdgpu gpurun_out/r5j/wgrad_sweep.txt | cut -c1-120 | tail -14
