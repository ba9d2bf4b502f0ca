#!/bin/bash
# Pipelined AND sequential ms per step of the headline for several builds of the library, alternating, on one box:
#     tools/ab_libs.sh libmccnn_hip.so libX.so ...        (variants: MCCNN_LIB_NAME=libX.so python -m mccnn_amd.build, then touch)
# A kernel that is faster alone can make the pipelined step slower (it runs beside the convolution kernels): the
# op-level time of a geometry kernel is not enough to accept a change.
run() { MCCNN_LIB_NAME=$1 python bench.py --no-configs --no-layers --no-cpu-baseline --no-breakdown --scaling weak 2>/dev/null | python -c "import sys,json; d=[l for l in sys.stdin if l.startswith('details: ')][-1]; d=json.loads(d[9:]); print('$1', 'pipelined', d['ms_per_step'], 'sequential', d['config']['sequential_ms_per_step'])"; }
for k in 1 2 3; do for l in "$@"; do run $l; done; done
