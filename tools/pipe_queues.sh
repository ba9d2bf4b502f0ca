#!/bin/bash
# tools/pipe_queues.sh <tag> [cfgs...]: kernel trace of the PIPELINED step of each configuration -> per-queue view
# (tools/queues.py) + launch sequence of one step (tools/step_trace.py) -> gpurun_out/pipe_<tag>/<cfg>_{queues,step}.txt
set -u
TAG=$1; shift
CFGS=${*:-"cfg2 cfg3 cfg4"}
REPO=$PWD
OUT=$REPO/gpurun_out/pipe_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for cfg in $CFGS; do
    rm -rf $OUT/trace_$cfg
    PIPE=1 rocprofv3 --kernel-trace -d $OUT/trace_$cfg -o t --output-format csv -- python $REPO/tools/config_time.py $cfg 20 > $OUT/$cfg.log 2>&1
    python $REPO/tools/queues.py $OUT/trace_$cfg 0.3 > $OUT/${cfg}_queues.txt 2>&1
    python $REPO/tools/step_trace.py $OUT/trace_$cfg 20 > $OUT/${cfg}_step.txt 2>&1
    tail -n 1 $OUT/$cfg.log
    rm -rf $OUT/trace_$cfg
done
