"""Host run-ahead of a configuration's pipelined steps: unbounded, or at most one / two steps (an event wait per step): python tools/lag_probe.py cfgN"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from mccnn_amd.workloads import CONFIGS
torch.cuda.set_device(0)
torch.autograd.set_multithreading_enabled(False)
cw = bench.ConfigWorkload(CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"], torch.device("cuda", 0))
cw.set_pipeline(True, geometry=True)
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
for lag in (0, 1, 2, 0, 1, 2):
    cw.lag = lag
    ms, _ = cw.timed(30, 5)
    print(name, "lag", lag, "%.3f ms" % ms, flush=True)
