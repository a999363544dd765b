#!/bin/bash
# A/B of two library builds on the weight-gradient kernels (conv micro-bench, same box, interleaved)
OUT=gpurun_out/$1; mkdir -p $OUT
for rep in 1 2; do
for lib in base new; do
  if [ $lib = base ]; then export SEMSEG_NATIVE_LIB=$PWD/build_ab/libsemseg_base.so; else unset SEMSEG_NATIVE_LIB; fi
  echo "== $lib (rep $rep)"
  timeout 120 python tools/conv_bench.py --mode h2 --passes wgrad --verify --iters 10 --layers stem_conv2,stem_conv3,l1_conv2 --tile 1 --split 56 2>&1 | grep -v "^sum" | cut -c1-120
  timeout 120 python tools/conv_bench.py --mode h2 --passes wgrad --verify --iters 10 --layers l2_conv2,l3_conv2_d2,hr_48,hr_96,hr_192,l4_conv2_d4,l3_conv3,l4_conv1,deepsup 2>&1 | grep -v "^sum" | cut -c1-120
done; done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
