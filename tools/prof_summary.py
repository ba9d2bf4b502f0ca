"""Summarise rocprofv3 kernel-trace stats + PMC passes written by tools/prof.sh."""
import glob
import os
import sys

import pandas as pd

out = sys.argv[1]


def short(n):
    n = n.split("(")[0]
    return n.replace("void mccnn::", "").replace("mccnn::", "")[:60]


for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    df = pd.read_csv(f)
    df["Name"] = df["Name"].map(short)
    cols = [c for c in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage") if c in df.columns]
    print("== kernel stats (%s)" % os.path.relpath(f, out))
    print(df[cols].head(25).to_string(index=False))
rows = []
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    df = pd.read_csv(f)
    df["Kernel_Name"] = df["Kernel_Name"].map(short)
    g = df.groupby(["Kernel_Name", "Counter_Name"])["Counter_Value"].mean().reset_index()
    rows.append(g)
if rows:
    allc = pd.concat(rows)
    piv = allc.pivot_table(index="Kernel_Name", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
    keep = [k for k in piv.index if any(t in k for t in ("conv_", "f1_", "neigh", "pdf_edges", "edge_rec", "scatter_edge", "keys_hist"))]
    pd.set_option("display.width", 250)
    pd.set_option("display.max_columns", 50)
    print("== PMC (mean per dispatch)")
    print(piv.loc[keep].T.to_string(float_format=lambda x: "%.4g" % x))

# HBM-side traffic per dispatch of the conv kernels, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for
# gfx950 (FETCH_SIZE counts 64 B per 128-B request: doubled; both counters are in KB; WRITE_SIZE is uncalibrated).
if rows:
    import json
    traffic = {}
    for k in piv.index:
        if "FETCH_SIZE" in piv.columns and "WRITE_SIZE" in piv.columns and ("conv_" in k or "f1_" in k or "neigh" in k or "pdf_edges" in k):
            f, w = piv.loc[k].get("FETCH_SIZE"), piv.loc[k].get("WRITE_SIZE")
            if f == f and w == w:
                traffic[k] = {"fetch_bytes": float(f) * 1024 * 2, "write_bytes": float(w) * 1024,
                              "bytes": float(f) * 1024 * 2 + float(w) * 1024}
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
