#!/bin/bash
# forward op time / pipelined / sequential step of the headline for library builds: tools/f1fwd_ab.sh libA.so libB.so ...
run() { MCCNN_LIB_NAME=$1 python bench.py --no-configs --no-layers --no-cpu-baseline --scaling weak 2>/dev/null | python -c "import sys,json; d=[l for l in sys.stdin if l.startswith('details: ')][-1]; d=json.loads(d[9:]); print('$1', 'pipelined', d['ms_per_step'], 'sequential', d['config']['sequential_ms_per_step'], 'fwd', d['breakdown']['spatial_conv_fwd']['ms'])"; }
for k in 1 2; do for l in "$@"; do run $l; done; done
