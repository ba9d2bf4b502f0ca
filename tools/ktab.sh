#!/bin/bash
# Kernel-level A/B (rocprofv3 --kernel-trace --stats): tools/ktab.sh "<layers>" lib1 lib2 ...   (on the GPU box)
LAYERS=$1; shift
export TMPDIR=/tmp
REPO=$PWD
for lib in "$@"; do
  for L in $LAYERS; do
    OUT=/tmp/kt_${lib}_$L; rm -rf $OUT
    (cd /tmp && MCCNN_TORCH_EXT=${KTAB_TORCH_EXT:-1} MCCNN_LIB_NAME=$lib rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python $REPO/bench.py --steps 10 --warmup 2 --layer $L --no-cpu-baseline --no-configs --scaling weak --no-breakdown > /dev/null 2>&1)
    echo "== $lib $L"; python tools/kt.py $OUT
  done
done
