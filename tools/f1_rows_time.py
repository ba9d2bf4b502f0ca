"""1to64 forward on the 100k room: factored edge-streaming kernels (mccnn_spatial_conv_fwd) against the row-per-lane
edge pass (mccnn_spatial_conv_fwd_f1_rows): python tools/f1_rows_time.py [Fout]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mccnn_amd.MCConvModule as M  # noqa: E402
from mccnn_amd._lib import check, ptr, stream_handle  # noqa: E402
from mccnn_amd.workloads import make_room  # noqa: E402

fout = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R, W, B = 0.1, 0.2, 1
pts = make_room(100000, 20180601)
P = torch.from_numpy(pts).cuda()
Bi = torch.zeros((len(pts), 1), dtype=torch.int32, device="cuda")
mn, mx = M.compute_aabb(P, Bi, B, False)
keys, idx = M.sort_points_step1(P, Bi, mn, mx, B, R, False)
rng = np.random.default_rng(1)
feats = torch.from_numpy((2 * rng.random((len(pts), 1)) - 1).astype(np.float32)).cuda()
sP, sB, sF, cells = M.sort_points_step2(P, Bi, feats, keys, idx, mn, mx, B, R, False)
start, packed = M.find_neighbors(P, Bi, sP, cells, mn, mx, R, B, False)
pdfs = M.compute_pdf(sP, sB, mn, mx, start, packed, W, R, B, False)
n, m, e = sP.shape[0], P.shape[0], packed.shape[0]
nb = (fout + 7) // 8
g = torch.Generator(device="cuda").manual_seed(1)
w1, w2, w3 = (torch.rand(s, device="cuda", generator=g) - 0.5 for s in ((3, 8 * nb), (8, 8 * nb), (8, 8 * nb)))
b1, b2, b3 = (0.1 * (torch.rand(8 * nb, device="cuda", generator=g) - 0.5) for _ in range(3))
lib = M._lib.load()
args = (ptr(sP), ptr(sF), ptr(sB), ptr(pdfs), ptr(P), ptr(start), ptr(packed), ptr(mn), ptr(mx), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3))


def ev(fn, iters=20):
    fn()
    ts = []
    for _ in range(iters):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.mean(ts))


o1 = torch.empty((m, fout), device="cuda")
sb = lib.mccnn_spatial_conv_state_bytes(m, e, 1, fout, 1)
state = torch.empty(sb, dtype=torch.uint8, device="cuda")
ws = torch.empty(max(256, lib.mccnn_spatial_conv_fwd_workspace_bytes(m, e, 1, fout, 1)), dtype=torch.uint8, device="cuda")
t1 = ev(lambda: check(lib.mccnn_spatial_conv_fwd(*args, n, m, e, 1, fout, 1, B, R, 0, 1, ptr(o1), ptr(state), ptr(ws), ws.numel(), stream_handle()), "fwd"))
pl = M._row_plan(packed, False, sP, sB, pdfs, P, start, packed, mn, mx, n, m, e, B, R, False, True, centre_points=P)
o2 = torch.empty((m, fout), device="cuda")
cs = torch.empty(sb - ((e * 16 + 255) // 256 * 256), dtype=torch.uint8, device="cuda")
scr = torch.empty((pl.scratch_rows, 8 * nb + 4), device="cuda")
t2 = ev(lambda: check(lib.mccnn_spatial_conv_fwd_f1_rows(*args, n, m, e, fout, B, R, 0, 1, ptr(pl.vrow), ptr(pl.vcode), ptr(pl.slice_off), ptr(pl.vpos_row), ptr(pl.rec), ptr(pl.other), ptr(o2), ptr(cs), ptr(scr), stream_handle()), "fwd_rows"))
print("1to%d forward: streaming %.4f ms, rows %.4f ms, max rel diff %.2e" % (fout, t1, t2, float((o1 - o2).abs().max() / o1.abs().max())))
