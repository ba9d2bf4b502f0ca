"""cProfile of the pipelined bench step loop (100k-point room): where the Python time of a step goes."""
import argparse, cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ap = argparse.Namespace(points=100000, radius=0.1, window=0.2, layer='1to64', rooms_per_gpu=1, steps=20, warmup=5,
                        scaling='weak', strong_rooms=8, gpus=1, no_cpu_baseline=True, no_breakdown=True, no_layers=True,
                        no_pipeline=False)
wl = bench.Workload(ap, '1to64', [20180601], 0, 1, torch.device('cuda', 0))
for _ in range(30):
    wl.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    wl.step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
