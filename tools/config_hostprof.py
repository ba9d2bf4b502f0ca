"""Host profile (cProfile) of one BASELINE configuration's steps: python tools/config_hostprof.py cfg4 30"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mccnn_amd.workloads import CONFIGS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
torch.cuda.set_device(0)
torch.autograd.set_multithreading_enabled(False)
cw = bench.ConfigWorkload(CONFIGS[name], torch.device("cuda", 0))
if os.environ.get("PIPE") == "1":   # the training-loop step: hierarchy two ahead + next batch's geometry, host one step ahead
    cw.set_pipeline(True, geometry=True)
    cw.builder.hostStepsAhead_ = 1
for _ in range(5):
    cw.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    cw.step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(30)
