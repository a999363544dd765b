#!/bin/bash
# round 3, GPU call 9: step-level A/B of the 16-wave tile (tuner candidates 0..10 vs 0..14)
TAG=${1:-r3i}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
for v in old:SEMSEG_TUNE_TILES=0,1,2,3,4,5,6,7,8,9,10,SEMSEG_TUNE_CACHE=/tmp/plans_old.json new:SEMSEG_TUNE_CACHE=/tmp/plans_new.json old2:SEMSEG_TUNE_TILES=0,1,2,3,4,5,6,7,8,9,10,SEMSEG_TUNE_CACHE=/tmp/plans_old.json new2:SEMSEG_TUNE_CACHE=/tmp/plans_new.json; do
  name=${v%%:*}; kv=${v#*:}
  # SEMSEG_TUNE_TILES holds commas itself: split the spec on ",SEMSEG_"
  kv1=${kv%%,SEMSEG_TUNE_CACHE*}; kv2=SEMSEG_TUNE_CACHE${kv##*SEMSEG_TUNE_CACHE}
  if [ "$kv1" = "$kv" ]; then envs=("$kv"); else envs=("$kv1" "$kv2"); fi
  env "${envs[@]}" timeout 400 python bench.py --steps 40 --warmup 6 --no-cpu-baseline > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(python -c "import json;d=json.load(open('$OUT/ab_$name.json'));print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['plan_tile_split'])")"
done
cp /tmp/plans_new.json $OUT/plans_new.json
for c in 2 4; do
  echo "== bench config $c"
  SEMSEG_TUNE_CACHE=/tmp/plans_new_c$c.json timeout 900 python bench.py --config $c --steps 40 --warmup 8 --no-cpu-baseline > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; echo "rc=$?"; python -c "import json;d=json.load(open('$OUT/bench_cfg$c.json'));print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['plan_tile_split'])"
done
