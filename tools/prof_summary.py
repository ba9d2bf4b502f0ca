"""Summarise the rocprofv3 passes written by tools/prof.sh: python tools/prof_summary.py <dir> [drop]

Kernel times come from the per-dispatch kernel trace with the first `drop` dispatches of every kernel left out (steady
state); counters are means per dispatch. HBM-side traffic is corrected as /opt/skills/guides/MI355X_MICROARCH.md
prescribes for gfx950: FETCH_SIZE counts 64 B per 128-B request (doubled), both counters are in KB, WRITE_SIZE is
uncalibrated. Writes <dir>/traffic.json and <dir>/kernels.json next to the text summary."""
import glob
import json
import os
import sys


def _src_sha():
    # sha1 of the kernel sources these numbers were taken from (bench.kernel_sources_sha1 computes the same): a reader of a
    # committed profile -- and bench.py, which quotes its traffic -- can tell whether it belongs to the tree at hand
    import glob
    import hashlib
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mccnn_amd", "csrc")
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


SRC_SHA = _src_sha()

import pandas as pd

out = sys.argv[1]
drop = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def short(n):
    n = n.split("(")[0]
    return n.replace("void mccnn::", "").replace("mccnn::", "")[:64]


kern = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    df = pd.read_csv(f)
    df["Name"] = df["Kernel_Name"].map(short)
    df["ns"] = df["End_Timestamp"] - df["Start_Timestamp"]
    df = df.sort_values("Start_Timestamp")
    rows = []
    for name, g in df.groupby("Name", sort=False):
        steady = g["ns"].iloc[drop:] if len(g) > drop else g["ns"]
        rows.append((name, len(g), len(steady), steady.mean() / 1e3, steady.min() / 1e3, g["ns"].iloc[0] / 1e3, steady.sum() / 1e3,
                     int(g["VGPR_Count"].iloc[0]), int(g["Scratch_Size"].iloc[0]), int(g["LDS_Block_Size"].iloc[0])))
    t = pd.DataFrame(rows, columns=["kernel", "calls", "steady", "avg_us", "min_us", "first_us", "total_us", "vgpr", "scratch", "lds"])
    t = t.sort_values("total_us", ascending=False)
    t["pct"] = 100 * t["total_us"] / t["total_us"].sum()
    print("== kernels, steady state (first %d dispatches of each kernel dropped; %s)" % (drop, os.path.relpath(f, out)))
    pd.set_option("display.width", 250)
    print(t.head(40).to_string(index=False, float_format=lambda x: "%.1f" % x))
    kern = {r.kernel: {"avg_us": round(r.avg_us, 2), "calls": int(r.calls), "vgpr": r.vgpr, "scratch": r.scratch} for r in t.itertuples()}
rows = []
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    df = pd.read_csv(f)
    df["Kernel_Name"] = df["Kernel_Name"].map(short)
    g = df.groupby(["Kernel_Name", "Counter_Name"])["Counter_Value"].mean().reset_index()
    rows.append(g)
if rows:
    allc = pd.concat(rows)
    piv = allc.pivot_table(index="Kernel_Name", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
    tags = ("conv_", "f1_", "neigh", "pdf_", "edge_rec", "scatter_edge", "keys_", "grid_", "sort_", "cell_", "poisson", "tr_", "scan", "rank_", "move_", "dw_", "sell_", "rows_", "vr_", "plan_", "reduce_", "permute_", "park_", "aabb_")
    keep = [k for k in piv.index if any(t in k for t in tags)]
    pd.set_option("display.max_columns", 60)
    pd.set_option("display.max_rows", 200)
    print("== PMC (mean per dispatch)")
    print(piv.loc[keep].T.to_string(float_format=lambda x: "%.4g" % x))
    # derived: MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in piv.columns and "GRBM_GUI_ACTIVE" in piv.columns:
        print("== MFMA pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs))")
        for k in keep:
            a, b = piv.loc[k].get("SQ_VALU_MFMA_BUSY_CYCLES"), piv.loc[k].get("GRBM_GUI_ACTIVE")
            if a == a and b == b and b > 0 and a > 0:
                print("   %-50s %.3f" % (k, a / (b / 8 * 1024)))
                kern.setdefault(k, {})["mfma_busy"] = round(a / (b / 8 * 1024), 4)
    traffic = {}
    for k in piv.index:
        if "FETCH_SIZE" in piv.columns and "WRITE_SIZE" in piv.columns and k in keep:
            f, w = piv.loc[k].get("FETCH_SIZE"), piv.loc[k].get("WRITE_SIZE")
            if f == f and w == w:
                traffic[k] = {"fetch_bytes": float(f) * 1024 * 2, "write_bytes": float(w) * 1024,
                              "bytes": float(f) * 1024 * 2 + float(w) * 1024}
    traffic["_kernel_sources_sha1"] = SRC_SHA
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
kern["_kernel_sources_sha1"] = SRC_SHA
json.dump(kern, open(os.path.join(out, "kernels.json"), "w"), indent=1)
