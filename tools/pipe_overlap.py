"""Pipelined vs sequential steps in ONE rocprofv3 kernel trace of `python bench.py --no-layers --no-breakdown
--no-cpu-baseline`: per-kernel mean duration in the two regions, the step period, and the timeline of one pipelined
step (main queue, with the side queue's kernels beside it).
    cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt -o t --output-format csv -- python $REPO/bench.py --steps 40 ...
    python tools/pipe_overlap.py /tmp/kt/t_kernel_trace.csv"""
import sys

import numpy as np
import pandas as pd

df = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp").reset_index(drop=True)
df["n"] = df.Kernel_Name.str.replace("void mccnn::", "").str.replace("mccnn::", "").str.split("(").str[0].str[:30]
df["dur"] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
queues = df.groupby("Queue_Id").size().sort_values(ascending=False)
main_q, side_q = queues.index[0], queues.index[1]
side = df[df.Queue_Id == side_q]
tp0, tp1 = side.Start_Timestamp.min(), side.End_Timestamp.max()
pipe = df[(df.Start_Timestamp >= tp0) & (df.End_Timestamp <= tp1)]
seq = df[df.Start_Timestamp > tp1]
a = pipe.groupby("n").dur.agg(["mean", "count"])
b = seq.groupby("n").dur.agg(["mean", "count"])
t = a.join(b, lsuffix="_pipelined", rsuffix="_sequential", how="outer").sort_values("mean_pipelined", ascending=False)
print("== mean kernel duration (us): pipelined region (geometry of batch k+1 on the side queue) vs sequential region")
print(t.round(1).to_string())
m = pipe[pipe.Queue_Id == main_q].reset_index(drop=True)
fw = m[m.n.str.startswith("f1_fwd_edges")]
if len(fw) > 12:
    st = fw.Start_Timestamp.values
    print("\nstep period, pipelined region (profiler attached): %.1f us" % (np.diff(st)[5:-2].mean() / 1e3))
    i0, i1 = fw.index[len(fw) // 2], fw.index[len(fw) // 2 + 1]
    t0, t1 = m.iloc[i0].Start_Timestamp, m.iloc[i1].Start_Timestamp
    print("\n== one pipelined step: main queue (start us, duration, gap before)")
    prev = None
    for i in range(i0, i1 + 1):
        r = m.iloc[i]
        gap = (r.Start_Timestamp - prev) / 1e3 if prev else 0.0
        print("%8.1f  dur %7.1f  gap %6.1f  %s" % ((r.Start_Timestamp - t0) / 1e3, r.dur, gap, r.n))
        prev = r.End_Timestamp
    print("== the same window, side queue")
    sd = pipe[(pipe.Queue_Id == side_q) & (pipe.Start_Timestamp >= t0) & (pipe.Start_Timestamp < t1)]
    for _, r in sd.iterrows():
        print("%8.1f  dur %7.1f  %s" % ((r.Start_Timestamp - t0) / 1e3, r.dur, r.n))
