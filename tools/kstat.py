import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
import bench
from mccnn_amd.workloads import CONFIGS
torch.cuda.set_device(0)
cw = bench.ConfigWorkload(CONFIGS["cfg2"], torch.device("cuda", 0))
cw.step()
for key, (st, pk) in cw.builder.cacheNeighs_.items():
    s = st.reshape(-1).cpu().numpy().astype(np.int64)
    e = pk.shape[0]
    k = np.diff(np.concatenate([s, [e]]))
    print(key[:60], "m", len(k), "e", e, "mean", k.mean().round(1), "max", k.max(), "p99", np.percentile(k, 99), "sum k^2 (M)", round((k * k).sum() / 1e6, 1), ">192:", int((k > 192).sum()))
