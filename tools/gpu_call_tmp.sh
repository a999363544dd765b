OUT=gpurun_out/r6k; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail $OUT/build.log; exit 1; }
timeout 900 python -m pytest tests -m gpu -q -x -k "every_tile_pinned or epilogue_statistics" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for tile in 15 19 20 21 16 18; do for sp in 0 1; do
  SEMSEG_DMA64_SPREAD=$sp timeout 200 python tools/conv_bench.py --mode h2 --layers hr_48,hr_96,hr_192,hr_384,l2_conv2,l1_conv2,l3_conv2_d2,l3_conv1 --passes fwd,dgrad --tile $tile --iters 40 > $OUT/conv_t${tile}_s${sp}.txt 2>&1
  echo "tile $tile spread $sp: $(grep -a 'default' $OUT/conv_t${tile}_s${sp}.txt | awk '{printf "%s %s %s | ", $1,$2,$6}' | cut -c1-700)"
done; done
b() { name=$1; cfg=$2; shift; shift; env "$@" timeout 400 python bench.py --config $cfg --steps 30 --warmup 6 --no-cpu-baseline --no-other-configs --no-box --no-scaling-model --repeats 1 > $OUT/bench_$name.json 2> $OUT/bench_$name.err; python - $OUT/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], d['ms_per_step'],'ms', d['config']['repeat_windows']['ms_per_step'])
except Exception as e: print(sys.argv[2],'no line',e)
PY
}
b h0 4 SEMSEG_DMA64_SPREAD=0; b h1 4 SEMSEG_DMA64_SPREAD=1; b h0b 4 SEMSEG_DMA64_SPREAD=0; b h1b 4 SEMSEG_DMA64_SPREAD=1
b c0 1 SEMSEG_DMA64_SPREAD=0; b c1 1 SEMSEG_DMA64_SPREAD=1; b c0b 1 SEMSEG_DMA64_SPREAD=0; b c1b 1 SEMSEG_DMA64_SPREAD=1
