bash tools/gpu_run.sh r5p "test:(every_tile_pinned or epilogue_statistics) and 24"
timeout 900 python tools/conv_bench.py --mode h2 --passes fwd,dgrad --sweep --verify --layers l3_conv2_d2,l3_conv3,l3_conv1,l4_conv3,l4_conv1,l4_down,l4_conv2_d4,deepsup,conv_last > gpurun_out/r5p/fwd_dgrad_sweep.txt 2>&1
grep -v amdgpu gpurun_out/r5p/fwd_dgrad_sweep.txt | python -c "
import sys,re
for ln in sys.stdin:
    if ' default ' not in ln or 'sum over' in ln: continue
    cf=re.findall(r't(\d+)_s(\d+):(-?\d+)', ln)
    best=max((int(tf),int(t),int(s)) for t,s,tf in cf)
    b24=max(((int(tf),int(t),int(s)) for t,s,tf in cf if int(t)==24), default=None)
    b22=max(((int(tf),int(t),int(s)) for t,s,tf in cf if int(t)==22), default=None)
    print(ln.split()[0], ln.split()[1], 'best', best, 't22', b22, 't24', b24)
"
