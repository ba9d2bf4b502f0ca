"""Print the kernel timeline of the last bench step from a rocprofv3 kernel trace: start offset, duration, gap before."""
import glob
import sys

import pandas as pd

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
df = pd.read_csv(f).sort_values("Start_Timestamp")
names = df["Kernel_Name"].str.replace("void mccnn::", "").str.replace("mccnn::", "").str.split("(").str[0].str[:44]
# last step = from the last aabb_init on
idx = [i for i, n in enumerate(names) if n.startswith("aabb_init") or n.startswith("keys_hist")]
start = idx[-1]
t0 = df["Start_Timestamp"].iloc[start]
prev_end = t0
for i in range(start, len(df)):
    s, e = df["Start_Timestamp"].iloc[i], df["End_Timestamp"].iloc[i]
    print("%8.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, names.iloc[i]))
    prev_end = e
