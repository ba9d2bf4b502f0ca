"""The launch sequence of ONE steady-state step out of a rocprofv3 kernel trace:
    python tools/step_trace.py <trace dir> <steps in the trace> [which step from the end, default 2]
Prints, in start order, every dispatch of that step: offset from the step's first kernel, duration, gap to the previous
kernel's end on the same queue, queue, grid / workgroup size, name. Used to count and read the launch chain of a network
step (what is a memset, what is a scan, what depends on what)."""
import glob
import sys

import pandas as pd

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
steps = int(sys.argv[2])
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
df = pd.read_csv(f).sort_values("Start_Timestamp").reset_index(drop=True)
df["n"] = df.Kernel_Name.str.replace("void mccnn::", "").str.replace("mccnn::", "").str.split("(").str[0].str[:40]
# a step's marker: the kernel name with exactly `steps + warm-up` regular occurrences -- take aabb_reduce when present,
# otherwise the rarest kernel; the step = [marker[-back-1], marker[-back])
cnt = df.n.value_counts()
marker = None
for cand in ("aabb_reduce", "aabb_one", "aabb_points", "aabb_all"):
    if cand in cnt.index:
        marker = cand
        break
if marker is None:
    marker = cnt.index[-1]
idx = df.index[df.n == marker].tolist()
per = max(1, round(len(idx) / max(1, steps + 5)))
idx = idx[::per]
a, b = idx[-back - 1], idx[-back]
g = df.iloc[a:b].copy()
t0 = g.Start_Timestamp.min()
last_end = {}
print("marker %s (%d per step), step of %d dispatches, %.1f us from first start to last end, sum of kernel times %.1f us" % (
    marker, per, len(g), (g.End_Timestamp.max() - t0) / 1e3, (g.End_Timestamp - g.Start_Timestamp).sum() / 1e3))
for _, r in g.iterrows():
    q = r.Queue_Id
    gap = (r.Start_Timestamp - last_end[q]) / 1e3 if q in last_end else float("nan")
    last_end[q] = r.End_Timestamp
    print("%9.1f %7.1f gap %6.1f q%-2s grid %8d wg %4d  %s" % ((r.Start_Timestamp - t0) / 1e3, (r.End_Timestamp - r.Start_Timestamp) / 1e3, gap, q,
                                                         r.Grid_Size_X if "Grid_Size_X" in r else r.Grid_Size, r.Workgroup_Size_X if "Workgroup_Size_X" in r else r.Workgroup_Size, r.n))
print("\nper kernel in this step:")
t = g.assign(dur=(g.End_Timestamp - g.Start_Timestamp) / 1e3).groupby("n").dur.agg(["count", "sum", "mean"]).sort_values("sum", ascending=False)
print(t.round(1).to_string())
