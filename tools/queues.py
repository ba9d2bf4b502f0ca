"""Per-queue view of a rocprofv3 kernel trace: python tools/queues.py <dir> [skip_fraction]
For each HIP queue: kernels, busy time; the union of busy intervals against the wall time of the window (GPU idle = the
host is the bound); per queue the kernels by total time. The first `skip_fraction` (default 0.4) of the trace is dropped
(warm-up)."""
import glob
import sys

import numpy as np
import pandas as pd

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
df = pd.read_csv(f).sort_values("Start_Timestamp").reset_index(drop=True)
df["n"] = df.Kernel_Name.str.replace("void mccnn::", "").str.replace("mccnn::", "").str.split("(").str[0].str[:34]
t0, t1 = df.Start_Timestamp.min(), df.End_Timestamp.max()
w0 = t0 + (t1 - t0) * skip
df = df[df.Start_Timestamp >= w0]
wall = (df.End_Timestamp.max() - df.Start_Timestamp.min()) / 1e3
iv = df[["Start_Timestamp", "End_Timestamp"]].values
busy, cur_s, cur_e = 0.0, iv[0][0], iv[0][1]
for s_, e_ in iv[1:]:
    if s_ > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
busy += cur_e - cur_s
print("window %.1f us, some queue busy %.1f us (%.1f %%), kernels %d, sum of kernel times %.1f us" % (
    wall, busy / 1e3, busy / 10 / wall, len(df), ((df.End_Timestamp - df.Start_Timestamp).sum()) / 1e3))
for q, g in sorted(df.groupby("Queue_Id"), key=lambda kv: -len(kv[1])):
    d = (g.End_Timestamp - g.Start_Timestamp) / 1e3
    print("\nqueue %s: %d kernels, busy %.1f us (%.1f %% of the window)" % (q, len(g), d.sum(), d.sum() / wall * 100))
    t = g.assign(dur=d).groupby("n").dur.agg(["sum", "count", "mean"]).sort_values("sum", ascending=False).head(12)
    print(t.round(1).to_string())
