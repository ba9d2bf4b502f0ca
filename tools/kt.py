import sys, pandas as pd, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
df = pd.read_csv(f)
for _, r in df.iterrows():
    n = r["Name"]
    if any(k in n for k in ("dw_", "conv_bwd", "edge_records", "reduce_partials", "scatter_edge", "conv_fwd", "conv_stream", "tr_", "f1_", "neigh", "pdf_", "scan_", "pad_points")):
        print("   %-50s %8.1f us" % (n.split("(")[0].replace("void mccnn::", "")[:50], r["AverageNs"] / 1e3))
