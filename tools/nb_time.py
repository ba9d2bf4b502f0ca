"""find_neighbors on the 100k room: ms per call (count + scan + fill, HIP events) -- for A/B builds (MCCNN_NW_G ...)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, argparse
from mccnn_amd import MCConvModule as M
ap = argparse.Namespace(points=100000, radius=0.1, window=0.2, layer='1to64', rooms_per_gpu=int(os.environ.get("ROOMS", "1")), steps=20, warmup=5,
                        scaling='weak', strong_rooms=8, gpus=1, no_cpu_baseline=True, no_breakdown=True, no_layers=True, no_pipeline=True)
seeds = [20180601 + r for r in range(ap.rooms_per_gpu)]
wl = bench.Workload(ap, '1to64', seeds, 0, 1, torch.device('cuda', 0))
P, Bi, B = wl.P, wl.Bi, wl.B
mn, mx = wl.ph.aabbMin_, wl.ph.aabbMax_
keys, idx = M.sort_points_step1(P, Bi, mn, mx, B, 0.1, False)
sP, sB, sF, cells = M.sort_points_step2(P, Bi, wl.F.detach(), keys, idx, mn, mx, B, 0.1, False)
for _ in range(10):
    st, pk = M.find_neighbors(P, Bi, sP, cells, mn, mx, 0.1, B, False)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    st, pk = M.find_neighbors(P, Bi, sP, cells, mn, mx, 0.1, B, False)
e1.record(); torch.cuda.synchronize()
print("find_neighbors ms %.4f  E %d" % (e0.elapsed_time(e1) / 50, pk.shape[0]))
# order-sensitive checksum of the list (A/B runs of two fill passes must print the same value)
w = torch.arange(1, pk.shape[0] + 1, device=pk.device, dtype=torch.int64)
print("checksum", int(((pk[:, 0].to(torch.int64) * 31 + pk[:, 1].to(torch.int64)) * (w % 1000003)).sum()), int(st.to(torch.int64).sum()))
