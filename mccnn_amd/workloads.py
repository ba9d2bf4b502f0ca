"""Synthetic inputs and convolution graphs of the BASELINE.json configurations.

The reference ships no datasets and no benchmark inputs; what it fixes are the SHAPES: which point hierarchy a network
builds and which convolutions (levels, radii, KDE windows, feature counts) it runs over it. This module holds those
tables -- every row cites the create_convolution call it mirrors -- and the seeded generators of the point clouds the
tests and bench.py feed them with. Host-side NumPy only; nothing here touches the GPU.

  cfg0  one 4 096-point uniform cloud, one same-level convolution 3 -> 8          (BASELINE.json configs[0])
  cfg1  MCClassS, 32 clouds x 1 024 points, grow 16                                (models/MCClassS.py:29-71)
  cfg2  MCClassH, 32 clouds x 4 096 points, 3 Poisson levels, both logit branches  (models/MCClassH.py:30-187)
  cfg3  MCSeg, 16 clouds x 8 192 points, grow 32, bf16 feature rows (extension)    (models/MCSeg.py:29-198)
  cfg4  MCSegScanNet, ~100 000-point rooms, grow 64, absolute radii                (models/MCSegScanNet.py:29-249)
"""
import math
from collections import namedtuple

import numpy as np

S3 = math.sqrt(3.0) + 0.1

#: one create_convolution call: name, input level, output level, radius, KDE window, Fin, Fout, multiFeatureConv,
#: bf16 feature rows (extension; depth-wise layers of cfg3 only)
Conv = namedtuple("Conv", "name lin lout radius window fin fout combin bf16")


def _c(name, lin, lout, radius, window, fin, fout, combin, bf16=False):
    return Conv(name, lin, lout, radius, window, fin, fout, combin, bf16)


def mcclass_s(k):
    """models/MCClassS.py:38-71 (ConvolutionBuilder(KDEWindow=0.2))."""
    return [_c("Conv_1", 0, 1, 0.2, 0.2, 1, k, True),
            _c("Conv_2", 1, 2, 0.8, 0.2, 2 * k, 2 * k, False),
            _c("Conv_3", 2, 3, S3, 0.2, 4 * k, 4 * k, False)]


def mcclass_h(k):
    """models/MCClassH.py:40-187: both logit branches (ConvolutionBuilder(KDEWindow=0.25), pooling layers 0.2)."""
    return [_c("Conv_1", 0, 0, 0.1, 0.25, 1, k, True),
            _c("Pool_1", 0, 1, 0.2, 0.2, 2 * k, 2 * k, False),
            _c("Conv_2", 1, 1, 0.4, 0.25, 2 * k, 2 * k, False),
            _c("Pool_2", 1, 2, 0.8, 0.2, 8 * k, 8 * k, False),
            _c("Conv_3", 2, 2, 1.2, 0.25, 8 * k, 8 * k, False),
            _c("Pool_3", 2, 3, S3, 0.2, 32 * k, 32 * k, False),
            _c("Conv_2_2", 1, 1, 0.4, 0.25, 1, 2 * k, True),
            _c("Pool_2_2", 1, 2, 0.8, 0.2, 8 * k, 8 * k, False),
            _c("Conv_3_2", 2, 2, 1.2, 0.25, 8 * k, 8 * k, False),
            _c("Pool_3_2", 2, 3, S3, 0.2, 32 * k, 32 * k, False)]


def mcseg(k, bf16=True):
    """models/MCSeg.py:36-198: encoder, decoder and the skip up-samplings."""
    d = lambda *a: _c(*a, False, bf16)
    return [_c("Conv_1", 0, 0, 0.03, 0.25, 1, k, True),
            d("Pool_1", 0, 1, 0.05, 0.2, 2 * k, 2 * k),
            d("Conv_2", 1, 1, 0.1, 0.25, 2 * k, 2 * k),
            d("Pool_2", 1, 2, 0.2, 0.2, 4 * k, 4 * k),
            d("Conv_3", 2, 2, 0.4, 0.25, 4 * k, 4 * k),
            d("Pool_3", 2, 3, 0.8, 0.2, 8 * k, 8 * k),
            d("Conv_4", 3, 3, S3, 0.25, 8 * k, 8 * k),
            d("Up_3_4", 3, 2, S3, 0.25, 16 * k, 16 * k),
            d("DeConv_3", 2, 2, 0.4, 0.25, 8 * k, 8 * k),
            d("Up_2_3", 2, 1, 0.2, 0.25, 8 * k, 8 * k),
            d("DeConv_2", 1, 1, 0.1, 0.25, 4 * k, 4 * k),
            d("Up_1_2", 1, 0, 0.05, 0.25, 4 * k, 4 * k),
            d("Up_1_3", 2, 0, 0.2, 0.25, 8 * k, 8 * k),
            d("DeConv_1", 0, 0, 0.03, 0.25, 4 * k, 4 * k)]


def mcseg_scannet(k):
    """models/MCSegScanNet.py:37-244 (ConvolutionBuilder(KDEWindow=0.25, relativeRadius=False), hierarchy radii
    [0.1, 0.2, 0.4, 0.8] absolute)."""
    d = lambda name, lin, lout, r, w, f: _c(name, lin, lout, r, w, f, f, False)
    return [_c("Pool_0", 0, 1, 0.1, 0.2, 1, k, True),
            d("Conv_1", 1, 1, 0.4, 0.25, k),
            d("Pool_1", 1, 2, 0.4, 0.2, 2 * k),
            d("Conv_2", 2, 2, 0.8, 0.25, 2 * k),
            d("Pool_2", 2, 3, 0.8, 0.2, 4 * k),
            d("Conv_3", 3, 3, 1.6, 0.25, 4 * k),
            d("Pool_3", 3, 4, 1.6, 0.2, 8 * k),
            d("Conv_4", 4, 4, 5.0, 0.25, 8 * k),
            d("Up_3_4", 4, 3, 1.6, 0.25, 8 * k),
            d("DeConv_3", 3, 3, 1.6, 0.25, 8 * k),
            d("Up_2_3", 3, 2, 0.8, 0.25, 4 * k),
            d("DeConv_2", 2, 2, 0.8, 0.25, 4 * k),
            d("Up_1_2", 2, 1, 0.4, 0.25, 2 * k),
            d("Up_1_3", 3, 1, 0.8, 0.25, 2 * k),
            d("Up_1_4", 4, 1, 1.6, 0.25, 2 * k),
            d("DeConv_1", 1, 1, 0.4, 0.25, 4 * k),
            d("Up_0_1", 1, 0, 0.1, 0.25, 4 * k)]


Config = namedtuple("Config", "name what clouds points relative hierarchy convs cloud_kind seed")

CONFIGS = {
    "cfg0": Config("cfg0", "one 4096-point uniform cloud, one same-level convolution 3->8, relative radius 0.1",
                   1, 4096, True, [], [_c("Conv", 0, 0, 0.1, 0.2, 3, 8, True)], "uniform", 1),
    "cfg1": Config("cfg1", "MCClassS graph, 32 clouds x 1024 points, grow 16", 32, 1024, True, [0.1, 0.4, S3],
                   mcclass_s(16), "modelnet", 41),
    "cfg2": Config("cfg2", "MCClassH graph (both branches), 32 clouds x 4096 points, grow 16", 32, 4096, True,
                   [0.1, 0.4, S3], mcclass_h(16), "modelnet", 43),
    "cfg3": Config("cfg3", "MCSeg graph, 16 clouds x 8192 points, grow 32, bf16 feature rows in the depth-wise layers",
                   16, 8192, True, [0.025, 0.1, 0.4], mcseg(32), "modelnet", 47),
    "cfg4": Config("cfg4", "MCSegScanNet graph, 100000-point non-uniform room(s), grow 64, absolute radii", 1, 100000,
                   False, [0.1, 0.2, 0.4, 0.8], mcseg_scannet(64), "room", 20180601),
}


def conv_nb(fin, fout, combin):
    """Blocks of 8 neurons of the kernel MLP (MCConvBuilder.py:396-399)."""
    neurons = fin * fout if combin else fin
    return (neurons + 7) // 8


# ---------------------------------------------------------------------------------------------- point clouds
def make_room(n, seed, oversample=3.0):
    """Synthetic ScanNet-like room: surfaces of a 6.0 x 4.0 x 2.8 m box (floor + 4 walls)
    plus 6 axis-aligned furniture boxes, sampled uniformly by area, then thinned to n points with the
    reference's *gradient* protocol along the longest axis (utils/DataSet.py:431-492:
    keep-prob = sqrt(clip((x - 0.2 L) / (0.6 L), 0.01, 1)))."""
    rng = np.random.default_rng(seed)
    L, W, H = 6.0, 4.0, 2.8
    rects = []  # (origin, edge u, edge v)
    rects.append(((0, 0, 0), (L, 0, 0), (0, W, 0)))  # floor
    rects.append(((0, 0, 0), (L, 0, 0), (0, 0, H)))
    rects.append(((0, W, 0), (L, 0, 0), (0, 0, H)))
    rects.append(((0, 0, 0), (0, W, 0), (0, 0, H)))
    rects.append(((L, 0, 0), (0, W, 0), (0, 0, H)))
    frng = np.random.default_rng(20180601)  # furniture layout is fixed across rooms
    for _ in range(6):
        sx, sy, sz = 0.4 + 1.2 * frng.random(), 0.4 + 0.8 * frng.random(), 0.3 + 0.9 * frng.random()
        ox, oy = (L - sx) * frng.random(), (W - sy) * frng.random()
        o = np.array([ox, oy, 0.0])
        for (a, u, v) in (((0, 0, sz), (sx, 0, 0), (0, sy, 0)), ((0, 0, 0), (sx, 0, 0), (0, 0, sz)),
                          ((0, sy, 0), (sx, 0, 0), (0, 0, sz)), ((0, 0, 0), (0, sy, 0), (0, 0, sz)),
                          ((sx, 0, 0), (0, sy, 0), (0, 0, sz))):
            rects.append((tuple(o + np.array(a)), u, v))
    areas = np.array([np.linalg.norm(np.cross(u, v)) for (_, u, v) in rects])
    total = int(n * oversample)
    which = rng.choice(len(rects), size=total, p=areas / areas.sum())
    uv = rng.random((total, 2))
    org = np.array([r[0] for r in rects], dtype=np.float64)[which]
    eu = np.array([r[1] for r in rects], dtype=np.float64)[which]
    ev = np.array([r[2] for r in rects], dtype=np.float64)[which]
    p = org + eu * uv[:, :1] + ev * uv[:, 1:]
    prob = np.sqrt(np.clip((p[:, 0] - 0.2 * L) / (0.6 * L), 0.01, 1.0))
    keep = rng.random(total) < prob
    p = p[keep]
    if len(p) < n:
        return make_room(n, seed, oversample * 1.6)
    sel = rng.choice(len(p), size=n, replace=False)
    return p[sel].astype(np.float32)


def modelnet_like(n, B, seed):
    """B synthetic shapes with n points each, normalised like the ModelNet loader (centred, inside the unit sphere):
    an ellipsoid shell plus the faces of a box, different proportions per cloud."""
    rng = np.random.default_rng(seed)
    pts, bids = [], []
    for b in range(B):
        k = n // 2
        v = rng.normal(size=(k, 3))
        shell = v / np.linalg.norm(v, axis=1, keepdims=True) * (0.3 + 0.7 * rng.random(3))
        box = (rng.random((n - k, 3)) - 0.5) * (0.4 + 1.2 * rng.random(3))
        face = rng.integers(0, 3, n - k)
        half = (np.abs(box).max(axis=0) + 1e-3)
        box[np.arange(n - k), face] = np.sign(box[np.arange(n - k), face]) * half[face]
        p = np.concatenate([shell, box])
        p -= p.mean(axis=0)
        p /= np.linalg.norm(p, axis=1).max()
        rng.shuffle(p)
        pts.append(p)
        bids.append(np.full((n, 1), b, np.int32))
    return np.concatenate(pts).astype(np.float32), np.concatenate(bids)


def uniform_cloud(n, seed):
    """cfg0: n points ~ U[0,1)^3, one cloud."""
    rng = np.random.default_rng(seed)
    return rng.random((n, 3), dtype=np.float32), np.zeros((n, 1), np.int32)


def rooms(n, seeds):
    """A batch of rooms (cfg4: one per GPU; several per GPU in the strong-scaling runs)."""
    pts = np.concatenate([make_room(n, s) for s in seeds])
    bids = np.repeat(np.arange(len(seeds), dtype=np.int32), n).reshape(-1, 1)
    return pts, bids


def config_points(cfg, clouds=None, first_cloud=0):
    """(points [N,3] f32, batch ids [N,1] i32, number of clouds) of a configuration; `clouds` / `first_cloud` select a
    sub-batch (data-parallel shards, the bounded CPU sample)."""
    B = cfg.clouds if clouds is None else clouds
    if cfg.cloud_kind == "uniform":
        p, b = uniform_cloud(cfg.points, cfg.seed)
        return p, b, 1
    if cfg.cloud_kind == "room":
        p, b = rooms(cfg.points, [cfg.seed + first_cloud + r for r in range(B)])
        return p, b, B
    p, b = modelnet_like(cfg.points, first_cloud + B, cfg.seed)
    lo = first_cloud * cfg.points
    return p[lo:], b[lo:] - first_cloud, B
