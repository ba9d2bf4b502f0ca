#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/prof.sh <tag> <bench args...>
# Kernel-trace stats pass + separate PMC passes (never combined with other trace domains).
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
CMD="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d $OUT/pmc1 -o p --output-format csv -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS -d $OUT/pmc2 -o p --output-format csv -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_LEVEL_WAVES -d $OUT/pmc3 -o p --output-format csv -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc4 -o p --output-format csv -- $CMD > $OUT/pmc4.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc5 -o p --output-format csv -- $CMD > $OUT/pmc5.log 2>&1
cd $REPO
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
