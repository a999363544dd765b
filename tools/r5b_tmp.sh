mkdir -p gpurun_out/r5e
python __graft_entry__.py > gpurun_out/r5e/build.log 2>&1
timeout 300 python tools/probes/winograd_dgrad_pass.py --time --probes --iters 10 --geom 2,64,64,4096,512,1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5e/probes.txt
