"""Poisson sampling kernels on the finest levels of BASELINE cfg3 (16 x 8192 points, relative radius 0.025) and cfg4
(100k room, absolute radius 0.1): ms per call of poisson_sampling (count + fill, one read-back)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mccnn_amd.MCConvModule as M  # noqa: E402
from mccnn_amd.workloads import CONFIGS, config_points  # noqa: E402

for name in ("cfg1", "cfg3", "cfg4"):
    cfg = CONFIGS[name]
    pts, bids, B = config_points(cfg)
    P, Bi = torch.from_numpy(pts).cuda(), torch.from_numpy(bids).cuda()
    F = torch.ones((len(pts), 1), device="cuda")
    r = cfg.hierarchy[0]
    mn, mx = M.compute_aabb(P, Bi, B, cfg.relative)
    k, i = M.sort_points_step1(P, Bi, mn, mx, B, r, cfg.relative)
    sP, sB, sF, c = M.sort_points_step2(P, Bi, F, k, i, mn, mx, B, r, cfg.relative)
    ts = []
    for it in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = M.poisson_sampling(sP, sB, c, mn, mx, r, B, cfg.relative)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%s level 1: %d points -> %d samples, cells %s, poisson_sampling %.3f ms (min of 8)" % (name, len(pts), out[0].shape[0], tuple(c.shape[:4]), min(ts)))
