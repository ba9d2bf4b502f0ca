#!/bin/bash
# Usage (on the GPU box, from the repo root):
#     tools/prof.sh <tag> <bench args...>            profile `python bench.py <args>` (headline layer only)
#     PROF_CMD="python tools/hier_time.py" tools/prof.sh <tag>      profile another command
# The profiled steps are strictly sequential (--no-pipeline): kernel times and counters of one kernel at a time.
# One kernel-trace stats pass + SEPARATE PMC passes (counters are never combined with other trace domains).
# Averages are taken over the steady state: prof_summary.py drops the first WARM dispatches of every kernel (the cold
# call that pages code objects in is 2-5x slower than the rest and used to bias "AverageNs" of a 3-step run).
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
STEPS=${PROF_STEPS:-20}
WARM=${PROF_WARM:-5}
CMD=${PROF_CMD:-"python $REPO/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-breakdown --no-layers --no-configs --scaling weak --no-pipeline $*"}
echo "$CMD" > $OUT/command.txt
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- $CMD > $OUT/trace.log 2>&1
if [ -z "${PROF_NO_PMC:-}" ]; then
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d $OUT/pmc1 -o p --output-format csv -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS -d $OUT/pmc2 -o p --output-format csv -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_LEVEL_WAVES -d $OUT/pmc3 -o p --output-format csv -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc4 -o p --output-format csv -- $CMD > $OUT/pmc4.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc5 -o p --output-format csv -- $CMD > $OUT/pmc5.log 2>&1
fi
cd $REPO
python tools/prof_summary.py $OUT ${PROF_WARM_DROP:-$((WARM + 1))} > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
