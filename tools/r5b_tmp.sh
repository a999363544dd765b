mkdir -p gpurun_out/r5d
python __graft_entry__.py > gpurun_out/r5d/build.log 2>&1
for g in 2,64,64,4096,512,1 2,128,128,2048,512,1 2,64,64,1024,512,1; do
  timeout 300 python tools/probes/winograd_dgrad_pass.py --time --iters 10 --geom $g 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r5d/dgrad_forms.txt
