"""find_neighbors kernel times on the 100k room against a UNIFORM cloud with the same mean row length: is the search bound
by its densest cells? python tools/nb_uniform.py (prints per-kernel HIP-event times of count / fill via the C-ABI)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mccnn_amd import MCConvModule as M  # noqa: E402
from mccnn_amd.workloads import make_room  # noqa: E402


def run(name, pts, r):
    P = torch.from_numpy(pts).cuda()
    Bi = torch.zeros((len(pts), 1), dtype=torch.int32, device="cuda")
    mn, mx = M.compute_aabb(P, Bi, 1, False)
    sP, sB, cells, idx, inv = M.build_grid(P, Bi, mn, mx, 1, r, False)
    for _ in range(5):
        st, pk = M.find_neighbors(P, Bi, sP, cells, mn, mx, r, 1, False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        st, pk = M.find_neighbors(P, Bi, sP, cells, mn, mx, r, 1, False)
    b.record()
    torch.cuda.synchronize()
    k = np.diff(np.append(st.reshape(-1).cpu().numpy(), pk.shape[0]))
    occ = (cells[..., 1] - cells[..., 0]).reshape(-1).cpu().numpy()
    print("%-8s find_neighbors %.4f ms  E %d  row length mean %.1f max %d  points per occupied cell mean %.1f max %d" % (
        name, a.elapsed_time(b) / 50, pk.shape[0], k.mean(), k.max(), occ[occ > 0].mean(), occ.max()))


room = make_room(100000, 20180601)
run("room", room, 0.1)
# uniform points in a slab whose volume gives the same mean row length (surfaces -> volume: only the density matters)
rng = np.random.default_rng(5)
vol = 100000 * (4.0 / 3.0) * np.pi * 0.1 ** 3 / 45.4
side = (vol / 0.3) ** 0.5
uni = (rng.random((100000, 3)) * np.array([side, side, 0.3])).astype(np.float32)
run("uniform", uni, 0.1)
