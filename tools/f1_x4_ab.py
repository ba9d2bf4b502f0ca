"""Forward pass of combin layers with one input feature on the 100k room: 64-edge chunks against four edges per lane
(mccnn_debug_f1_x4_min_edges), per number of MLP blocks: python tools/f1_x4_ab.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mccnn_amd import MCConvModule as mc, _lib  # noqa: E402
from mccnn_amd.workloads import make_room  # noqa: E402
from tests.helpers import make_mlp  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda", 0)
from mccnn_amd.workloads import CONFIGS, config_points  # noqa: E402
for n_pts, radius in ((100000, 0.1), (100000, 0.06), (400000, 0.1), ("cfg2", 0.1), ("cfg3", 0.03)):
    rel, B = False, 1
    if isinstance(n_pts, str):  # a batch of small clouds, radius relative to each cloud's box
        pts, bids, B = config_points(CONFIGS[n_pts])
        rel = True
    else:
        pts = (make_room(100000, 7) if n_pts == 100000 else np.concatenate([make_room(100000, 7 + k) + np.float32(k * 20.0) for k in range(4)])).astype(np.float32)
        bids = np.zeros((len(pts), 1), np.int32)
    P, Bi = torch.from_numpy(pts).to(dev), torch.from_numpy(bids).to(dev)
    mn, mx = mc.compute_aabb(P, Bi, B, rel)
    sP, sB, cells, idx, inv = mc.build_grid(P, Bi, mn, mx, B, radius, rel)
    C, Cb = (P, Bi) if os.environ.get('UNSORTED_CENTRES') else (sP, sB)  # the builder's centres are the level's points as given
    start, packed = mc.find_neighbors(C, Cb, sP, cells, mn, mx, radius, B, rel)
    pdfs = mc.compute_pdf(sP, sB, mn, mx, start, packed, 0.25, radius, B, rel)
    F = torch.rand((len(pts), 1), device=dev)
    for fout in (16, 32, 64):
        nb = (fout + 7) // 8
        w = make_mlp(nb, 3)
        ws = [torch.from_numpy(w[k]).to(dev) for k in ("w1", "w2", "w3", "b1", "b2", "b3")]
        res = []
        for min_e in (2 ** 31 - 1, 0):
            prev = lib.mccnn_debug_f1_x4_min_edges(min_e)
            for _ in range(3):
                mc.spatial_conv(sP, F, sB, pdfs, C, start, packed, mn, mx, *ws, fout, True, B, radius, rel, True)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                mc.spatial_conv(sP, F, sB, pdfs, C, start, packed, mn, mx, *ws, fout, True, B, radius, rel, True)
            b.record()
            b.synchronize()
            res.append(a.elapsed_time(b) / 10)
            lib.mccnn_debug_f1_x4_min_edges(prev)
        print("%d points, %d edges, %d blocks: chunks %.4f ms, four per lane %.4f ms" % (len(pts), packed.shape[0], nb, res[0], res[1]))
