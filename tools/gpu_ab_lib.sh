#!/bin/bash
# In-box A/B of two builds of libsemseg_hip.so on ANY probe command (interleaved, two repetitions):
#   cp semantic-segmentation-pytorch_amd/mit_semseg/_native/libsemseg_hip.so build_ab/libsemseg_base.so    # before the change; *.so is git-ignored but travels
#   gpurun --timeout 300 -- 'bash tools/gpu_ab_lib.sh <tag> build_ab/libsemseg_base.so python tools/conv_bench.py --mode h2 --passes wgrad --verify --layers stem_conv2'
# (whole-step A/B: tools/gpu_run.sh <tag> lib:build_ab/libsemseg_base.so quick ..., or its ab: stage with SEMSEG_NATIVE_LIB and a tune cache per side)
TAG=$1; BASE=$PWD/$2; shift 2; OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do
  for side in base new; do
    echo "== $side (rep $rep)"
    if [ $side = base ]; then SEMSEG_NATIVE_LIB=$BASE timeout 200 "$@"; else timeout 200 "$@"; fi 2>&1 | grep -v "amdgpu.ids" | cut -c1-220
  done
done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
