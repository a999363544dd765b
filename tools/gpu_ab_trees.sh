#!/bin/bash
# Round-over-round A/B INSIDE ONE BOX: the bench of an older tree (exported with `git archive <commit> | tar -x -C build_ab/<name>`) against
# the bench of this tree, interleaved:   gpurun --timeout 1200 -- 'bash tools/gpu_ab_trees.sh <tag> build_ab/r3_tree'
TAG=$1; OLD=$PWD/$2; ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( cd $OLD && python __graft_entry__.py > $OUT/build_old.log 2>&1 ) || { echo OLD BUILD FAILED; tail -5 $OUT/build_old.log; exit 1; }
python __graft_entry__.py > $OUT/build_new.log 2>&1 || { echo NEW BUILD FAILED; exit 1; }
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(d['ms_per_step'], 'ms', d['value'], 'img/s', 'roofline', d['roofline']['achieved'])
except Exception as e:
    print('no JSON line:', e)
PY
}
for rep in 1 2 3; do
  ( cd $OLD && SEMSEG_TUNE_CACHE=/tmp/plans_old.json timeout 600 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-other-configs > $OUT/old_$rep.json 2> $OUT/old_$rep.err ); echo "old tree  rep $rep: $(line $OUT/old_$rep.json)"
  timeout 600 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-other-configs --repeats 0 --no-scaling-model > $OUT/new_$rep.json 2> $OUT/new_$rep.err; echo "this tree rep $rep: $(line $OUT/new_$rep.json)"
done | tee $OUT/ab.txt
python - $OUT/new_3.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print('box:', d.get('box'))
PY
