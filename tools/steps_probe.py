"""ms/step of consecutive blocks of 20 steps in ONE process (the bench workload): first block after set-up, later blocks,
after torch.cuda.empty_cache() and after 2 s of idle -- python tools/steps_probe.py (on the GPU box)."""
import sys, time, argparse
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ap = argparse.Namespace(points=100000, radius=0.1, window=0.2, layer='1to64', rooms_per_gpu=1, steps=20, warmup=5,
                        scaling='weak', strong_rooms=8, gpus=1, no_cpu_baseline=True, no_breakdown=True, no_layers=True)
wl = bench.Workload(ap, '1to64', [20180601], 0, 1, torch.device('cuda', 0))
for rep in range(6):
    ms, val, _ = wl.timed(20, 4)
    print('block of 20 after 4 warm-up:', round(ms, 4))
ms, val, _ = wl.timed(100, 4)
print('block of 100:', round(ms, 4))
# per-step wall without sync between (host time to return)
torch.cuda.synchronize()
ts = []
t0 = time.perf_counter()
for i in range(30):
    wl.step()
    ts.append(time.perf_counter())
torch.cuda.synchronize()
print('per-step host return deltas (us):', [int((ts[i] - (ts[i-1] if i else t0)) * 1e6) for i in range(30)])
for rep in range(2):
    torch.cuda.empty_cache()
    ms, val, _ = wl.timed(20, 4)
    print('block of 20 after empty_cache + 4 warm-up:', round(ms, 4))
    ms, val, _ = wl.timed(20, 4)
    print('block of 20 again:', round(ms, 4))
time.sleep(2.0)
ms, val, _ = wl.timed(20, 4)
print('block of 20 after 2 s idle + 4 warm-up:', round(ms, 4))
ms, val, _ = wl.timed(20, 4)
print('block of 20 again:', round(ms, 4))
