#!/bin/bash
# Per-kernel averages of any command under rocprofv3: tools/ks.sh <tag> <command...>   (on the GPU box; output in gpurun_out/ks_<tag>.txt)
TAG=$1; shift
export TMPDIR=/tmp
REPO=$PWD
OUT=/tmp/ks_$TAG; rm -rf $OUT
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- "$@" > /tmp/ks_$TAG.log 2>&1)
mkdir -p $REPO/gpurun_out
python - "$OUT" <<'PY' | tee $REPO/gpurun_out/ks_$TAG.txt
import sys, glob, pandas as pd
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
d = pd.read_csv(f)
d["n"] = d.Name.str.replace("void mccnn::", "").str.replace("mccnn::", "").str.split("(").str[0].str.slice(0, 48)
d["avg_us"] = d.AverageNs / 1e3
d["tot_us"] = d.TotalDurationNs / 1e3
print(d[["n", "Calls", "avg_us", "tot_us"]].head(45).to_string())
PY
