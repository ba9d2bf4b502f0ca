#!/bin/bash
# A/B inside one box: single-workgroup small-problem kernels on / off, shortest plan pieces 4 / 16
for c in ${*:-cfg1 cfg2 cfg3 cfg4}; do
  echo "== $c: default | MCCNN_DEBUG=small_off | MCCNN_DEBUG=plan_min_l=16"
  python tools/config_time.py $c 40
  MCCNN_DEBUG=small_off python tools/config_time.py $c 40
  MCCNN_DEBUG=plan_min_l=16 python tools/config_time.py $c 40
done
