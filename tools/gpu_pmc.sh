#!/bin/bash
# PMC passes (separate runs, as MI355X_MICROARCH.md prescribes) on the conv microbench.
#   gpurun --timeout 900 -- 'bash tools/gpu_pmc.sh <tag> <layers> <passes>'
#   MODE=h2 TILE=5 SPLIT=4 bash tools/gpu_pmc.sh ...   (forced tile / split of the measured configuration)
TAG=${1:-pmc}; LAYERS=${2:-conv_last}; PASSES=${3:-fwd}; MODE=${MODE:-f32}; TILE=${TILE:--1}; SPLIT=${SPLIT:-0}
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
cd /tmp
CMD="python $ROOT/tools/conv_bench.py --mode $MODE --layers $LAYERS --passes $PASSES --iters 3 --tile $TILE --split $SPLIT"
run() { # name counters...
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o $n -- $CMD > $OUT/$n.log 2>&1
  echo "pmc pass $n rc=$?"
}
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
cd $ROOT
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
find $OUT -name '*.csv' -size +8M -delete
