#!/bin/bash
# A/B: run bench for each lib variant given; prints conv timings. Usage: tools/ab.sh "<layers>" lib1 lib2 ...
LAYERS=$1; shift
for lib in "$@"; do
  for L in $LAYERS; do
    MCCNN_LIB_NAME=$lib timeout 300 python bench.py --steps 5 --warmup 2 --layer $L --no-cpu-baseline --no-configs --scaling weak 2>/dev/null | python -c "import sys,json; d=[l for l in sys.stdin if l.startswith('details: ')][-1]; r=json.loads(d[9:]); b=r['breakdown']; print('$lib', r['config']['layer'], 'step', r['ms_per_step'], 'fwd', b['spatial_conv_fwd']['ms'], 'bwd', b['spatial_conv_bwd']['ms'])"
  done
done
