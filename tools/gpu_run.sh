#!/bin/bash
# ONE entry point for every GPU-box call of a round (replaces the per-call scripts of earlier rounds):
#   gpurun --timeout 1200 -- 'bash tools/gpu_run.sh <tag> <stage> [<stage> ...]'
# Output goes to gpurun_out/<tag>/ (merged back by gpurun).  Stages run in the order given:
#   bench            default bench command, exactly as the driver runs it (full JSON line incl. cpu_baseline + other_configs)
#   quick            bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-other-configs (twice: a, b)
#   cfg:N            bench.py --config N --steps 40 --warmup 8 --no-cpu-baseline
#   ab:NAME:K=V,...  quick bench with the given environment (A/B inside one box; repeat the stage for interleaved runs)
#   abc:N:NAME:K=V,...  the same for bench.py --config N
#   lib:PATH         quick bench against another build of the library (SEMSEG_NATIVE_LIB)
#   smoke            __graft_entry__.smoke()
#   tests            pytest -m gpu (whole suite)        tests-fast: without the tile-pinned sweep
#   test:EXPR        pytest -m gpu -k EXPR
#   ddp2             bench.py --gpus 2 on the ONE GPU (SEMSEG_BENCH_DEVICE=0, gloo): the self-launch + N > 1 path end to end
#   infer            tools/bench_infer.py
#   prof             rocprofv3 --kernel-trace --stats of the quick bench -> kernel_stats*.csv, trace_gaps.txt
#   prof:N           the same for bench.py --config N
#   conv:ARGS        tools/conv_bench.py ARGS (comma -> space), e.g. conv:--mode,h2,--layers,layer4_d4,--sweep
#   py:PATH+ARG+...  python PATH ARG ... (a probe under tools/probes; '+' separates the arguments)
TAG=${1:-run}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$PWD
python __graft_entry__.py > $OUT/build.log 2>&1 || { echo BUILD FAILED; tail -20 $OUT/build.log; exit 1; }
export SEMSEG_TUNE_CACHE=${SEMSEG_TUNE_CACHE:-/tmp/semseg_plans_h2.json}
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    oc = d['config'].get('other_configs') or {}
    print(d['ms_per_step'], 'ms', d['value'], 'img/s', 'roofline', d['roofline']['achieved'], d['roofline'].get('plan_tile_split'),
          'launch', str(d['config']['launch'])[:40], {k: v.get('img_s') for k, v in oc.items()})
except Exception as e:
    print('no JSON line:', e)
PY
}
q() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-other-configs > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name rc=$?: $(show $OUT/bench_$name.json)"; tail -2 $OUT/bench_$name.err | cut -c1-200; }
profile() { name=$1; shift
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_$name -o bench -- python $ROOT/bench.py --steps 10 --warmup 6 --no-cpu-baseline --no-other-configs --repeats 0 --no-box --no-scaling-model "$@" > $ROOT/$OUT/rocprof_$name.log 2>&1 ); echo "rocprof rc=$?"
  db=$(find $OUT/prof_$name -name '*.db' | head -1); tr=$(find $OUT/prof_$name -name '*kernel_trace.csv' | head -1); src=${db:-$tr}
  python tools/rocprof_summary.py $src $OUT/kernel_stats_$name.csv
  python tools/rocprof_summary.py $src $OUT/kernel_stats_by_grid_$name.csv --by-grid
  python tools/trace_gaps.py $src 0.6 > $OUT/trace_gaps_$name.txt; head -3 $OUT/trace_gaps_$name.txt
  ( cd tools && python trace_concurrency.py $ROOT/$src 8 > $ROOT/$OUT/trace_concurrency_$name.txt 2>&1 ); head -34 $OUT/trace_concurrency_$name.txt
  ( cd tools && python trace_context.py $ROOT/$src copyBuffer 0.5 > $ROOT/$OUT/trace_context_copybuffer_$name.txt 2>&1 ); head -14 $OUT/trace_context_copybuffer_$name.txt
  rm -rf $OUT/prof_$name; }
n=0
for stage in "$@"; do
  n=$((n+1)); echo "== [$n] $stage"
  case $stage in
    bench) ( unset SEMSEG_TUNE_CACHE; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ); echo "rc=$?: $(show $OUT/bench.json)"; cut -c1-400 $OUT/bench.json; tail -3 $OUT/bench.err | cut -c1-300 ;;
    quick) q a X=1; q b X=1; cp $SEMSEG_TUNE_CACHE $OUT/plans_h2.json 2>/dev/null ;;
    cfg:*) c=${stage#cfg:}; timeout 900 python bench.py --config $c --steps 40 --warmup 8 --no-cpu-baseline > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; echo "rc=$?: $(show $OUT/bench_cfg$c.json)"; tail -2 $OUT/bench_cfg$c.err | cut -c1-300 ;;
    abc:*) spec=${stage#abc:}; c=${spec%%:*}; spec=${spec#*:}; name=${spec%%:*}; kv=${spec#*:}; IFS=, read -ra kvs <<< "$kv"; env "${kvs[@]}" timeout 600 python bench.py --config $c --steps 40 --warmup 8 --no-cpu-baseline --no-other-configs --no-box --no-scaling-model > $OUT/bench_cfg${c}_${name}_$n.json 2> $OUT/bench_cfg${c}_${name}_$n.err; echo "cfg$c $name rc=$?: $(show $OUT/bench_cfg${c}_${name}_$n.json)"; tail -1 $OUT/bench_cfg${c}_${name}_$n.err | cut -c1-200 ;;
    ab:*) spec=${stage#ab:}; name=${spec%%:*}; kv=${spec#*:}; IFS=, read -ra kvs <<< "$kv"; q ${name}_$n "${kvs[@]}" ;;
    lib:*) q lib_$n SEMSEG_NATIVE_LIB=$ROOT/${stage#lib:} ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
    tests) timeout 2400 python -m pytest tests -m gpu -q -s --durations=80 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log | cut -c1-300; grep -a "FAILED\|Error" $OUT/pytest_gpu.log | cut -c1-300 | head -20 ;;
    tests-fast) timeout 1200 python -m pytest tests -m gpu -q -x -k "not every_tile_pinned" > $OUT/pytest_gpu_fast.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu_fast.log | cut -c1-300; grep -a "FAILED\|Error" $OUT/pytest_gpu_fast.log | cut -c1-300 | head -20 ;;
    test:*) timeout 1200 python -m pytest tests -m gpu -q -s -k "${stage#test:}" > $OUT/pytest_k_$n.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_k_$n.log | cut -c1-300 ;;
    ddp2) SEMSEG_BENCH_DEVICE=0 GPU_MAX_HW_QUEUES=2 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 > $OUT/bench_ddp2.json 2> $OUT/bench_ddp2.err; echo "rc=$?"; cut -c1-600 $OUT/bench_ddp2.json; tail -3 $OUT/bench_ddp2.err | cut -c1-300 ;;
    infer) timeout 300 python tools/bench_infer.py 2>&1 | grep images_per_sec | cut -c1-300 | tee $OUT/bench_infer.jsonl ;;
    prof) profile cfg1 ;;
    prof:*) profile cfg${stage#prof:} --config ${stage#prof:} ;;
    conv:*) args=${stage#conv:}; timeout 900 python tools/conv_bench.py ${args//,/ } > $OUT/conv_$n.txt 2>&1; echo "rc=$?"; tail -40 $OUT/conv_$n.txt | cut -c1-250 ;;
    py:*) a=${stage#py:}; timeout 900 python ${a//+/ } > $OUT/py_$n.txt 2>&1; echo "rc=$?"; tail -40 $OUT/py_$n.txt | cut -c1-250 ;;
    *) echo "unknown stage $stage" ;;
  esac
done
