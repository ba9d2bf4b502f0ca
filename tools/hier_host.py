"""Host time of PointHierarchy construction (cfg4 room), split by phase: python tools/hier_host.py [cfg4]"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from mccnn_amd.workloads import CONFIGS
import mccnn_amd.MCConvModule as M
from mccnn_amd import _lib
torch.cuda.set_device(0)
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
cw = bench.ConfigWorkload(CONFIGS[name], torch.device("cuda", 0))
lib = _lib.load()
orig = lib.mccnn_hierarchy_level
acc = {"c": 0.0, "n": 0}
def timed(*a):
    t = time.perf_counter(); r = orig(*a); acc["c"] += time.perf_counter() - t; acc["n"] += 1; return r
class L:  # proxy
    def __getattr__(self, k):
        return timed if k == "mccnn_hierarchy_level" else getattr(lib, k)
M._lib.load = lambda: L()
for _ in range(5): cw.hierarchy()
torch.cuda.synchronize()
acc["c"] = 0.0; acc["n"] = 0
w0 = M.host_wait_seconds()
t0 = time.perf_counter()
N = 50
for _ in range(N):
    cw.hierarchy()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("%s hierarchy: %.3f ms per build on the host, of which %.3f in %d mccnn_hierarchy_level calls, %.3f waiting for the sizes"
      % (name, (t1 - t0) / N * 1e3, acc["c"] / N * 1e3, acc["n"] // N, (M.host_wait_seconds() - w0) / N * 1e3))
