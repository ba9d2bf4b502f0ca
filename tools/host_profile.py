"""cProfile of a BASELINE.json configuration's step loop (host side): python tools/host_profile.py cfg4
Where the Python / autograd / ctypes time of a host-bound step goes (NOTES.md section 6b)."""
import os, sys, cProfile, pstats, torch
sys.path.insert(0, os.getcwd())
import bench
from mccnn_amd.workloads import CONFIGS
torch.cuda.set_device(0)
torch.autograd.set_multithreading_enabled(False)
name = sys.argv[1]
cw = bench.ConfigWorkload(CONFIGS[name], torch.device("cuda", 0))
if os.environ.get("PIPE", "1") == "1":
    print("pipelined:", cw.set_pipeline(True, geometry=os.environ.get("DEEP", "1") == "1"))
for _ in range(5): cw.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(30): cw.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumtime").print_stats(30)
