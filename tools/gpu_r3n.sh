#!/bin/bash
# round 3, GPU call 14: more tile forms (fwd/dgrad id 18: 128x128 on 4 waves + 3-slot ring; wgrad ids 8/9: 128x128 on 8 / 16 waves)
TAG=${1:-r3n}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
echo "== every tile pinned"
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "every_tile_pinned" > $OUT/pytest_tiles.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_tiles.log | cut -c1-300
grep -a "FAILED" $OUT/pytest_tiles.log | head -10 | cut -c1-200
echo "== wgrad sweep"
timeout 900 python tools/conv_bench.py --mode h2 --layers l4_conv3,l3_conv2_d2,l4_conv1,l3_conv3,l3_conv1,l2_conv2,l1_conv2,stem_conv2,stem_conv3,hr_48,hr_96,hr_192,hr_384,l1_conv3,l2_conv3 --passes wgrad --iters 8 --sweep > $OUT/wgrad_sweep.txt 2>&1; echo "rc=$?"
python - <<'PY'
import re
for line in open('gpurun_out/r3n/wgrad_sweep.txt'):
    if '| best' not in line: continue
    head, rest = line.split('| best',1)
    res = dict((m.group(1), float(m.group(2))) for m in re.finditer(r'(t\d+_s\d+):(-?\d+)', rest))
    old = max((v,k) for k,v in res.items() if int(k[1:].split('_')[0]) <= 7)
    new = max((v,k) for k,v in res.items() if int(k[1:].split('_')[0]) >= 8)
    print('%-36s best old %-8s %4.0f TF | new %-8s %4.0f TF' % (head[:36], old[1], old[0], new[1], new[0]))
PY
echo "== step A/B"
for cfg in 1 4; do
  for name in old new old2 new2; do
    case $name in old*) export SEMSEG_TUNE_TILES=0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17; export SEMSEG_TUNE_WTILES=0,1,2,3,4,5,6,7; export SEMSEG_TUNE_CACHE=/tmp/plans_old_c$cfg.json;; *) unset SEMSEG_TUNE_TILES; unset SEMSEG_TUNE_WTILES; export SEMSEG_TUNE_CACHE=/tmp/plans_new_c$cfg.json;; esac
    timeout 600 python bench.py --config $cfg --steps 30 --warmup 6 --no-cpu-baseline > $OUT/ab_c${cfg}_$name.json 2> $OUT/ab_c${cfg}_$name.err
    echo "cfg$cfg $name: $(python -c "import json;d=json.load(open('$OUT/ab_c${cfg}_$name.json'));print(d['ms_per_step'], d['value'])")"
  done
done
