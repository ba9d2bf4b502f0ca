"""How much of the geometry pass (sort, search, KDE) of the NEXT batch hides under the convolution kernels of the current
one when the two run on separate HIP streams: python tools/overlap_probe.py (on the GPU box)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from mccnn_amd import MCConvModule as M  # noqa: E402
from mccnn_amd._lib import ptr, stream_handle, check  # noqa: E402

ap = argparse.Namespace(points=100000, radius=0.1, window=0.2, layer='1to64', rooms_per_gpu=1, steps=20, warmup=5,
                        scaling='weak', strong_rooms=8, gpus=1, no_cpu_baseline=True, no_breakdown=True, no_layers=True)
dev = torch.device('cuda', 0)
wl = bench.Workload(ap, '1to64', [20180601], 0, 1, dev)
P, Bi, Fd, B, r, w = wl.P, wl.Bi, wl.F.detach(), wl.B, 0.1, 0.2
mn, mx = wl.ph.aabbMin_, wl.ph.aabbMax_
lib = M._lib.load()


def geom():
    keys, idx = M.sort_points_step1(P, Bi, mn, mx, B, r, False)
    sP, sB, sF, cells = M.sort_points_step2(P, Bi, Fd, keys, idx, mn, mx, B, r, False)
    start, packed = M.find_neighbors(P, Bi, sP, cells, mn, mx, r, B, False)
    pdfs = M.compute_pdf(sP, sB, mn, mx, start, packed, w, r, B, False)
    return sP, sB, sF, start, packed, pdfs, idx


g = geom()
sP, sB, sF, start, packed, pdfs, idx = g
ws_ = [p.detach() for p in wl.builder.parameters()]
w1, b1, w2, b2, w3, b3 = ws_[0], ws_[1], ws_[2].reshape(8, -1), ws_[3].reshape(-1), ws_[4].reshape(8, -1), ws_[5].reshape(-1)
n, m, e = sP.shape[0], P.shape[0], packed.shape[0]
fin, fout = 1, 64
o = torch.empty((m, fout), device=dev)
fwd_ws = torch.empty(max(256, lib.mccnn_spatial_conv_fwd_workspace_bytes(m, e, fin, fout, 1)), dtype=torch.uint8, device=dev)
bwd_ws = torch.empty(max(256, lib.mccnn_spatial_conv_bwd_workspace_bytes(n, m, e, fin, fout, 1)), dtype=torch.uint8, device=dev)
state = torch.empty(lib.mccnn_spatial_conv_state_bytes(m, e, fin, fout, 1), dtype=torch.uint8, device=dev)
fg = torch.empty_like(sF)
gws = [torch.empty_like(t) for t in (w1, b1, w2, b2, w3, b3)]
cargs = (ptr(sP), ptr(sF), ptr(sB), ptr(pdfs), ptr(P), ptr(start), ptr(packed), ptr(mn), ptr(mx), ptr(w1), ptr(b1), ptr(w2),
         ptr(b2), ptr(w3), ptr(b3))


def conv():
    check(lib.mccnn_spatial_conv_fwd(*cargs, n, m, e, fin, fout, 1, B, r, 0, 1, ptr(o), ptr(state), ptr(fwd_ws), fwd_ws.numel(),
                                     stream_handle()), "fwd")
    check(lib.mccnn_spatial_conv_bwd(*cargs, ptr(wl.OG), n, m, e, fin, fout, 1, B, r, 0, 1, ptr(state), None, None, ptr(fg),
                                     *[ptr(t) for t in gws], ptr(bwd_ws), bwd_ws.numel(), stream_handle()), "bwd")
    M._gather_rows(fg, idx, n)


def timeit(fn, iters=60):
    for _ in range(25):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


side = torch.cuda.Stream()


def both_sequential():
    geom()
    conv()


def both_overlapped():
    conv()                       # current batch, main stream (asynchronous)
    with torch.cuda.stream(side):
        geom()                   # next batch: the host waits for its edge count while the conv kernels run


print("geometry alone   ms", round(timeit(geom), 4))
print("conv alone       ms", round(timeit(conv), 4))
print("sequential       ms", round(timeit(both_sequential), 4))
print("two streams      ms", round(timeit(both_overlapped), 4))
