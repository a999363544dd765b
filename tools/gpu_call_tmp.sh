OUT=gpurun_out/r6V; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
LAY=l3_conv1,l3_conv2_d2,l4_conv1,l4_down,l2_conv2
for m in 1 10 11 12; do
  SEMSEG_DMA64_SPREAD=$m timeout 200 python tools/conv_bench.py --mode h2 --layers $LAY --passes fwd --sweep --tiles 22,23,24 --splits 1 --iters 20 > $OUT/bench_mode$m.txt 2>&1
done
paste -d'\n' $OUT/bench_mode1.txt $OUT/bench_mode10.txt $OUT/bench_mode11.txt $OUT/bench_mode12.txt | cut -c1-200
