#!/bin/bash
OUT=gpurun_out/$1; mkdir -p $OUT
L=l4_conv3,l4_conv1,l4_down,l3_conv3,l3_conv1,l1_conv1,l1_conv3,l2_conv3,cls,l3_conv2_d2,l2_conv2
for rep in 1 2; do
for o in 0 1; do
  echo "== order $o (rep $rep)"
  SEMSEG_IGEMM_ORDER=$o timeout 120 python tools/conv_bench.py --mode h2 --passes fwd,dgrad --verify --iters 20 --layers $L 2>&1 | grep -v "amdgpu.ids" | cut -c1-120
done; done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
