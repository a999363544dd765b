mkdir -p gpurun_out/r5h; python __graft_entry__.py > gpurun_out/r5h/build.log 2>&1
timeout 800 python tools/conv_bench.py --mode h2 --passes wgrad --sweep --verify --layers stem_conv2,stem_conv3,l1_conv2,l2_conv2,hr_48,hr_96,l3_conv2_d2 > gpurun_out/r5h/wgrad_sweep.txt 2>&1
grep -v "amdgpu.ids" gpurun_out/r5h/wgrad_sweep.txt | tail -60 | cut -c1-300
