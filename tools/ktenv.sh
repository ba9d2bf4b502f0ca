#!/bin/bash
# Kernel-level timing under rocprofv3 for the current environment: tools/ktenv.sh "<layers>" [tag]   (on the GPU box)
LAYERS=$1; TAG=${2:-cur}
export TMPDIR=/tmp
REPO=$PWD
for L in $LAYERS; do
  OUT=/tmp/kt_${TAG}_$L; rm -rf $OUT
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python $REPO/bench.py --steps 10 --warmup 2 --layer $L --no-cpu-baseline --no-configs --scaling weak --no-breakdown > /dev/null 2>&1)
  echo "== $TAG $L"; python tools/kt.py $OUT
done
