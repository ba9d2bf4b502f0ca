"""cProfile of the host path of ONE small convolution (cached geometry): python tools/host_micro.py [cfg4 Conv_3]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mccnn_amd.workloads import CONFIGS  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
layer = sys.argv[2] if len(sys.argv) > 2 else "Conv_3"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.autograd.set_multithreading_enabled(False)
cw = bench.ConfigWorkload(CONFIGS[cfg], dev)
cw.step()
ph = cw.ph
ci = [i for i, c in enumerate(CONFIGS[cfg].convs) if c.name == layer][0]
inputs = [cw.feats[ci]] + list(cw.builder.parameters())
N = 1000


def loop():
    for _ in range(N):
        o = cw.conv(ph, ci)
        torch.autograd.grad([o], inputs, [cw.ogs[ci]], allow_unused=True)


loop()
torch.cuda.synchronize()
t0 = time.perf_counter()
loop()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("forward + backward: %.1f us per call (host)" % ((t1 - t0) / N * 1e6))
pr = cProfile.Profile()
pr.enable()
loop()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
