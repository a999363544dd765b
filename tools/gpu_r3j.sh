#!/bin/bash
# round 3, GPU call 10: 16-wave weight-gradient tile: parity of every tile, sweep, step-level A/B
TAG=${1:-r3j}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
echo "== every tile pinned (wgrad) + winograd wgrad"
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "every_tile_pinned or winograd" > $OUT/pytest_tiles.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_tiles.log | cut -c1-300
grep -a "FAILED" $OUT/pytest_tiles.log | head -10 | cut -c1-200
echo "== wgrad sweep"
timeout 900 python tools/conv_bench.py --mode h2 --layers conv_last,deepsup,l4_conv2_d4,l4_conv3,l3_conv2_d2,l4_conv1,l4_down,l3_conv3,l3_conv1 --passes wgrad --iters 5 --sweep > $OUT/wgrad_sweep.txt 2>&1; echo "rc=$?"
python - <<'PY'
import re
for line in open('gpurun_out/r3j/wgrad_sweep.txt'):
    if '| best' not in line: continue
    head, rest = line.split('| best',1)
    res = dict((m.group(1), float(m.group(2))) for m in re.finditer(r'(t\d+_s\d+):(-?\d+)', rest))
    old = max((v,k) for k,v in res.items() if int(k[1:].split('_')[0]) <= 6)
    new = max((v,k) for k,v in res.items() if int(k[1:].split('_')[0]) >= 7)
    print('%-36s best old %-8s %4.0f TF | 16-wave %-8s %4.0f TF' % (head[:36], old[1], old[0], new[1], new[0]))
PY
echo "== step A/B: wgrad tiles 0..6 vs 0..7 (fwd/dgrad tiles 0..14 in both)"
export SEMSEG_TUNE_CACHE=/tmp/plans_fd.json
for name in old new old2 new2; do
  case $name in old*) export SEMSEG_TUNE_WTILES=0,1,2,3,4,5,6; export SEMSEG_TUNE_CACHE=/tmp/plans_wold.json;; *) unset SEMSEG_TUNE_WTILES; export SEMSEG_TUNE_CACHE=/tmp/plans_wnew.json;; esac
  timeout 400 python bench.py --steps 40 --warmup 6 --no-cpu-baseline > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(python -c "import json;d=json.load(open('$OUT/ab_$name.json'));print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['plan_tile_split'])")"
done
cp /tmp/plans_wnew.json $OUT/plans_new.json
