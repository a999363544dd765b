#!/bin/bash
OUT=gpurun_out/$1; mkdir -p $OUT
for rep in 1 2; do
for lib in base new; do
  if [ $lib = base ]; then export SEMSEG_NATIVE_LIB=$PWD/build_ab/libsemseg_base.so; else unset SEMSEG_NATIVE_LIB; fi
  echo "== $lib (rep $rep)"
  timeout 120 python tools/probes/winograd_midsize.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
  timeout 60 python tools/probes/winograd_dgrad_pass.py --iters 20 2>&1 | tail -4 | cut -c1-200
  timeout 60 python tools/probes/winograd_dgrad_pass.py --iters 20 --mode fwd --geom 2,64,64,4096,512,1 2>&1 | tail -4 | cut -c1-200
done; done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
