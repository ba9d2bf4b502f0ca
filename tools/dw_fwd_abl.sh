#!/bin/bash
# forward of dw256 on the room with ablation builds of conv_rows.hip (libabl1 = no gather, libabl2 = no MLP)
for lib in libmccnn_hip.so libabl1.so libabl2.so; do
  echo "== $lib"; MCCNN_LIB_NAME=$lib python tools/dw_time.py 256 2>&1 | grep "rows     "
done
