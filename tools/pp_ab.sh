#!/bin/bash
# pipelined-step timelines of two library builds: tools/pp_ab.sh libmccnn_hip.so libold.so
export TMPDIR=/tmp
R=$PWD
for l in "$@"; do
  rm -rf /tmp/kt_$l; mkdir -p $R/gpurun_out
  (cd /tmp && MCCNN_LIB_NAME=$l rocprofv3 --kernel-trace -d /tmp/kt_$l -o t --output-format csv -- python $R/bench.py --steps 40 --warmup 10 --no-configs --no-layers --no-cpu-baseline --no-breakdown --scaling weak > /dev/null 2>&1)
  echo "===================== $l"
  python $R/tools/pipe_overlap.py $(find /tmp/kt_$l -name t_kernel_trace.csv | head -1) 2>&1 | head -70
done
