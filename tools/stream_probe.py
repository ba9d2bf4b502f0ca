"""Probe of the run-ahead machinery on a NON-default stream: 300 pipelined steps of a BASELINE configuration inside
torch.cuda.stream(s), outputs compared bit for bit with a sequential step on the default stream; then 20 prefetch-and-drop steps
(nothing consumes what was started) and one more compared step.   python tools/stream_probe.py cfg2"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from mccnn_amd.workloads import CONFIGS
torch.cuda.set_device(0)
torch.autograd.set_multithreading_enabled(os.environ.get("AUTOGRAD_MT", "0") == "1")   # (AUTOGRAD_MT=1: the engine's device thread runs the backward passes)
name = sys.argv[1]
cw = bench.ConfigWorkload(CONFIGS[name], torch.device("cuda", 0))
# reference: sequential step on the default stream
ref = [o.detach().float().clone() for o in cw.step()]
torch.cuda.synchronize()
s = torch.cuda.Stream()
bad = 0
with torch.cuda.stream(s):
    assert cw.set_pipeline(True, geometry=True)
    for k in range(300):
        outs = cw.step()
        if k % 25 == 24:
            for a, b in zip(outs, ref):
                if not torch.equal(a.detach().float(), b):
                    bad += 1
    # prefetch and drop on the non-default stream
    for _ in range(20):
        cw.builder.reset()
        cw.ph = cw.ready_ph
        nxt = cw.hierarchy(cw.next_ph)
        cw.request_next()
        cw.builder.prefetch_step(nxt)
        cw.ready_ph = nxt
    outs = cw.step()
    for a, b in zip(outs, ref):
        if not torch.equal(a.detach().float(), b):
            bad += 1
s.synchronize()
torch.cuda.synchronize()
print(name, "non-default stream: mismatching outputs", bad)
