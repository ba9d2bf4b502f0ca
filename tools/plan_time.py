"""Row plans of the 100k room's neighbour list (ROOMS=n: n rooms in one batch), built from scratch REPS times: forward plan
(with the per-edge records) then the transposed list + transposed plan -- ms per pair (HIP events); under tools/ks.sh the
per-kernel averages.   python tools/plan_time.py [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mccnn_amd.MCConvModule as M  # noqa: E402
from mccnn_amd.workloads import make_room  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rooms = int(os.environ.get("ROOMS", "1"))
R, W = 0.1, 0.2
pts = np.concatenate([make_room(100000, 20180601 + r) for r in range(rooms)])
bid = np.concatenate([np.full((100000, 1), r, np.int32) for r in range(rooms)])
P = torch.from_numpy(pts).cuda()
Bi = torch.from_numpy(bid).cuda()
B = rooms
mn, mx = M.compute_aabb(P, Bi, B, False)
keys, idx = M.sort_points_step1(P, Bi, mn, mx, B, R, False)
feats = torch.zeros((len(pts), 8), device="cuda")
sP, sB, sF, cells = M.sort_points_step2(P, Bi, feats, keys, idx, mn, mx, B, R, False)
start, packed = M.find_neighbors(P, Bi, sP, cells, mn, mx, R, B, False)
pdfs = M.compute_pdf(sP, sB, mn, mx, start, packed, W, R, B, False)
n = m = len(pts)
e = packed.shape[0]
args = (sP, sB, pdfs, P, start, packed, mn, mx, n, m, e, B, R, False, True)


def once():
    for a in ("_mccnn_rowplans", "_mccnn_transposed", "_mccnn_transposed_event"):
        if hasattr(packed, a):
            delattr(packed, a)
    f = M._row_plan(packed, False, *args, centre_points=P)
    t = M._row_plan(packed, True, *args)
    return f, t


once()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    f, t = once()
e1.record()
torch.cuda.synchronize()
print("rooms %d  E %d: both plans %.3f ms per build (%d builds)" % (rooms, e, e0.elapsed_time(e1) / reps, reps))
