"""Host cost of a BASELINE configuration's step: python tools/host_floor.py [cfg1 cfg4 ...]

For every configuration: the wall time of the step loop up to the point where the LAST launch has been enqueued (host
time; the searches' edge-count waits are inside it) against the time after the final synchronisation, plus a host
micro-benchmark of one small convolution (op call with autograd / raw C-ABI call / allocations)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mccnn_amd import MCConvModule as M  # noqa: E402
from mccnn_amd.workloads import CONFIGS  # noqa: E402

names = sys.argv[1:] or ["cfg1", "cfg2", "cfg3", "cfg4"]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.autograd.set_multithreading_enabled(False)
for name in names:
    cw = bench.ConfigWorkload(CONFIGS[name], dev)
    for _ in range(5):
        cw.step()
    torch.cuda.synchronize()
    steps = 30
    waits = [0.0]
    orig = M._await_mailbox

    def timed_wait(view, _o=orig, _w=waits):
        t = time.perf_counter()
        r = _o(view)
        _w[0] += time.perf_counter() - t
        return r
    M._await_mailbox = timed_wait
    t0 = time.perf_counter()
    for _ in range(steps):
        cw.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    M._await_mailbox = orig
    print("%s: host %.3f ms per step (of which edge-count waits %.3f), after sync %.3f ms, launches %s" % (
        name, (t1 - t0) / steps * 1e3, waits[0] / steps * 1e3, (t2 - t0) / steps * 1e3,
        cw.timed(5, 0)[1]))

# micro-benchmark: the smallest depth-wise layer of cfg4 (host-bound by construction)
cw = bench.ConfigWorkload(CONFIGS["cfg4"], dev)
cw.step()
ph = cw.ph
ci = [i for i, c in enumerate(CONFIGS["cfg4"].convs) if c.name == "Conv_3"][0]
N = 300
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    o = cw.conv(ph, ci)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("create_convolution (cached geometry), forward only: %.1f us per call" % ((t1 - t0) / N * 1e6))
inputs = [cw.feats[ci]] + list(cw.builder.parameters())
t0 = time.perf_counter()
for _ in range(N):
    o = cw.conv(ph, ci)
    torch.autograd.grad([o], inputs, [cw.ogs[ci]], allow_unused=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("forward + backward: %.1f us per call" % ((t1 - t0) / N * 1e6))
t0 = time.perf_counter()
for _ in range(N):
    x = torch.empty((318, 256), dtype=torch.float32, device=dev)
t1 = time.perf_counter()
print("torch.empty: %.2f us" % ((t1 - t0) / N * 1e6))
lib = cw.lib
t0 = time.perf_counter()
for _ in range(N):
    lib.mccnn_debug_launch_count()
t1 = time.perf_counter()
print("ctypes call without arguments: %.2f us" % ((t1 - t0) / N * 1e6))
a = torch.zeros(16, device=dev)
t0 = time.perf_counter()
for _ in range(N):
    a.add_(1.0)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("torch in-place add (one launch): %.2f us" % ((t1 - t0) / N * 1e6))
