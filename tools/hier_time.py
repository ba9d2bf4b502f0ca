"""Wall time of a 4-level PointHierarchy of the 100k-point room (absolute radius), fused (one read-back) vs op by op
(one per level), and of one Poisson sampling: python tools/hier_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tests.helpers import make_room  # noqa: E402
import mccnn_amd.MCConvBuilder as MB  # noqa: E402
import mccnn_amd.MCConvModule as M  # noqa: E402

P = torch.from_numpy(make_room(100000, 20180601)).cuda()
Bi = torch.zeros((100000, 1), dtype=torch.int32, device="cuda")
F = torch.ones((100000, 1), device="cuda")
for fused in (True, False):
    MB.FUSED_HIERARCHY = fused
    ts = []
    for it in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ph = MB.PointHierarchy(P, F, Bi, [0.1, 0.2, 0.4, 0.8], "PH", 1, False)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("hierarchy %-10s ms (min of last 5) %.3f  sizes %s" % ("fused" if fused else "op-by-op", min(ts[3:]),
                                                                [int(p.shape[0]) for p in ph.points_]))
MB.FUSED_HIERARCHY = True
mn, mx = ph.aabbMin_, ph.aabbMax_
k, i = M.sort_points_step1(P, Bi, mn, mx, 1, 0.1, False)
sP, sB, sF, c = M.sort_points_step2(P, Bi, F, k, i, mn, mx, 1, 0.1, False)
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = M.poisson_sampling(sP, sB, c, mn, mx, 0.1, 1, False)
    torch.cuda.synchronize()
    print("poisson 100k r=0.1 ms", (time.perf_counter() - t0) * 1e3, out[0].shape[0])
