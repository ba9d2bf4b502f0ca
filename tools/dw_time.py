"""Depth-wise layers on the 100k room: row-per-lane kernels (conv_rows.hip) against the edge-streaming kernels, HIP-event
times of forward / backward with the plans cached, and the one-off cost of building the plans.
    python tools/dw_time.py [64 128 256]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mccnn_amd.MCConvModule as M  # noqa: E402
from mccnn_amd.workloads import make_room  # noqa: E402

torch.autograd.set_multithreading_enabled(False)
feats_list = [int(a) for a in sys.argv[1:]] or [64, 128, 256]
R, W, B = 0.1, 0.2, 1
pts = make_room(100000, 20180601)
P = torch.from_numpy(pts).cuda()
Bi = torch.zeros((len(pts), 1), dtype=torch.int32, device="cuda")
mn, mx = M.compute_aabb(P, Bi, B, False)
keys, idx = M.sort_points_step1(P, Bi, mn, mx, B, R, False)


def ev(fn, iters=10):
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


for F in feats_list:
    rng = np.random.default_rng(F)
    feats = torch.from_numpy((2 * rng.random((len(pts), F)) - 1).astype(np.float32)).cuda()
    og = torch.from_numpy((2 * rng.random((len(pts), F)) - 1).astype(np.float32)).cuda()
    sP, sB, sF, cells = M.sort_points_step2(P, Bi, feats, keys, idx, mn, mx, B, R, False)
    start, packed = M.find_neighbors(P, Bi, sP, cells, mn, mx, R, B, False)
    pdfs = M.compute_pdf(sP, sB, mn, mx, start, packed, W, R, B, False)
    nb = F // 8
    g = torch.Generator(device="cuda").manual_seed(1)
    w = [torch.rand(s, device="cuda", generator=g) - 0.5 for s in ((3, 8 * nb), (8, 8 * nb), (8, 8 * nb), (8 * nb,), (8 * nb,), (8 * nb,))]
    res = {}
    for rows in (True, False):
        M.ROW_KERNELS = rows
        sFr = sF.detach().clone().requires_grad_(True)
        ws_ = [t.clone().requires_grad_(True) for t in w]

        def fwd():
            return M.spatial_conv(sP, sFr, sB, pdfs, P, start, packed, mn, mx, ws_[0], ws_[1], ws_[2], ws_[3], ws_[4], ws_[5], F,
                                  False, B, R, False, True)
        t0 = time.perf_counter()
        out = fwd()
        torch.cuda.synchronize()
        t_first = (time.perf_counter() - t0) * 1e3
        t_f = ev(fwd)
        outs = [fwd() for _ in range(12)]
        it = iter(outs)
        t0 = time.perf_counter()
        gr = torch.autograd.grad([out], [sFr] + ws_, [og])
        torch.cuda.synchronize()
        t_bfirst = (time.perf_counter() - t0) * 1e3
        t_b = ev(lambda: torch.autograd.grad([next(it)], [sFr] + ws_, [og]))
        res[rows] = (out.detach(), [x.detach() for x in gr])
        print("F=%d %s: fwd %.3f ms (first call %.2f), bwd %.3f ms (first call %.2f)" % (
            F, "rows     " if rows else "streaming", t_f, t_first, t_b, t_bfirst))
    M.ROW_KERNELS = True
    a, b = res[True], res[False]
    rel = lambda x, y: float((x - y).abs().max() / y.abs().max())
    print("   rows vs streaming: out %.1e dFeat %.1e params %s" % (rel(a[0], b[0]), rel(a[1][0], b[1][0]),
                                                                  ["%.1e" % rel(x, y) for x, y in zip(a[1][1:], b[1][1:])]))
    pl = getattr(packed, "_mccnn_rowplans", {})
    for k, p in pl.items():
        if not isinstance(k, bool):
            continue
        S, o = p.num_slices, p.slice_off - p.buf.data_ptr()
        used = int(p.buf[o:o + 4 * (S + 1)].view(torch.int32)[S])
        print("   plan transposed=%s: %d slots used (%.3f x E), capacity %d" % (k, used, used / packed.shape[0], p.slot_capacity))
