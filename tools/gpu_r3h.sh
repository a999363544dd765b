#!/bin/bash
# round 3, GPU call 8: 4-wave / 16-wave forms of the LDS-DMA implicit-GEMM kernel: parity of every tile, sweep on the mid-size layers
TAG=${1:-r3h}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
echo "== every tile pinned"
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "every_tile_pinned" > $OUT/pytest_tiles.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_tiles.log | cut -c1-300
grep -a "FAILED" $OUT/pytest_tiles.log | head -10 | cut -c1-200
echo "== sweep"
timeout 900 python tools/conv_bench.py --mode h2 --layers conv_last,deepsup,l4_conv2_d4,l4_conv3,l3_conv2_d2,l4_conv1,l4_down,l3_conv3,l3_conv1 --passes fwd,dgrad --iters 5 --sweep > $OUT/conv_sweep.txt 2>&1; echo "rc=$?"
python - <<'PY'
import re
for line in open('gpurun_out/'+__import__('os').environ.get('TAGX','r3h')+'/conv_sweep.txt'):
    if '| best' not in line: continue
    head, rest = line.split('| best',1)
    res = dict((m.group(1), float(m.group(2))) for m in re.finditer(r'(t\d+_s\d+):(-?\d+)', rest))
    old = max((v,k) for k,v in res.items() if int(k[1:].split('_')[0]) <= 10)
    new = max((v,k) for k,v in res.items() if int(k[1:].split('_')[0]) >= 11)
    print('%-36s best old %-8s %4.0f TF | best new %-8s %4.0f TF' % (head[:36], old[1], old[0], new[1], new[0]))
PY
