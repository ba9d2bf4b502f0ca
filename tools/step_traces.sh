#!/bin/bash
# tools/step_traces.sh <tag> [cfgs...]: kernel trace of the strictly sequential step of each configuration + the launch
# sequence of one step (tools/step_trace.py) -> gpurun_out/steps_<tag>/<cfg>.txt
set -u
TAG=$1; shift
CFGS=${*:-"cfg1 cfg2 cfg3 cfg4"}
REPO=$PWD
OUT=$REPO/gpurun_out/steps_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for cfg in $CFGS; do
    rm -rf $OUT/trace_$cfg
    PIPE=${PIPE:-0} rocprofv3 --kernel-trace -d $OUT/trace_$cfg -o t --output-format csv -- python $REPO/tools/config_time.py $cfg 10 > $OUT/$cfg.log 2>&1
    python $REPO/tools/step_trace.py $OUT/trace_$cfg 10 > $OUT/$cfg.txt 2>&1
    tail -n 1 $OUT/$cfg.log
    rm -rf $OUT/trace_$cfg
done
