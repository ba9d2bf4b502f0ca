"""Host time of a BASELINE configuration's step, split into its own work and its waits for device-side sizes:
    python tools/host_busy.py cfg1 cfg4          (MCCNN_NATIVE=0 for the op-by-op builder path)"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from mccnn_amd.workloads import CONFIGS
torch.cuda.set_device(0)
torch.autograd.set_multithreading_enabled(False)
for name in (sys.argv[1:] or ["cfg0", "cfg1", "cfg2", "cfg3", "cfg4"]):
    cw = bench.ConfigWorkload(CONFIGS[name], torch.device("cuda", 0))
    best = None
    for rep in range(3):
        ms, launches = cw.timed(30, 5)
        if best is None or ms < best[0]:
            best = (ms, launches, cw.host_issue_ms, cw.host_wait_ms)
    ms, launches, issue, wait = best
    print("%s: %.3f ms/step, host issue %.3f (own work %.3f, waits %.3f), %d launches" % (name, ms, issue, issue - wait, wait, launches))
