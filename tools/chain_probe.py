"""The convolution chain of a BASELINE configuration (all layers forward, one autograd backward) with the geometry CACHED:
time until the host has issued the chain against time until the GPU has finished it.
    python tools/chain_probe.py cfg4 [cfg1 ...]
Is the launch floor of a step the host or the chain of dependent small kernels? (If the host is done long before the
GPU, a hipGraph replay of the same launches has nothing to remove. A torch.cuda.graph capture of the chain was tried: it
fails inside the extension -- waits on events recorded outside the capture.)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from mccnn_amd.workloads import CONFIGS

torch.cuda.set_device(0)
torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda", 0)
for name in (sys.argv[1:] or ["cfg4"]):
    cw = bench.ConfigWorkload(CONFIGS[name], dev)
    cw.step()
    cw.step()
    ph = cw.ph
    n = len(cw.cfg.convs)

    def chain():
        outs = [cw.conv(ph, ci) for ci in range(n)]
        return torch.autograd.grad(outs, cw.feats + cw.params, cw.ogs, allow_unused=True)

    for _ in range(5):
        chain()
    torch.cuda.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        chain()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_eager = time.perf_counter() - t0
    msg = "%s: %d layers, cached geometry: eager %.3f ms per chain (host issue %.3f)" % (name, n, t_eager / reps * 1e3, t_issue / reps * 1e3)
    print(msg, flush=True)
