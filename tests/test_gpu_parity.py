"""Parity of the HIP path (through the C-ABI) against the CPU oracle on identical seeded inputs.

Bar (BASELINE.json north_star): integer / index outputs bit-exact (sort keys and order, cell table,
neighbour lists, Poisson samples); float outputs within 1e-4 relative.
"""
import numpy as np
import pytest

from tests.helpers import make_cloud, make_room, make_mlp, conv_nb, run_chain, assert_float_close

pytestmark = pytest.mark.gpu

RTOL = 1e-4  # north_star: "fp32 features within 1e-4 rel"


def _wrap(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _unwrap(t):
    return t.detach().cpu().numpy()


def _ident(x):
    return x


def assert_close(got, ref, rtol=RTOL, what=""):
    """norm-wise (max |diff| / max |ref| <= rtol) and per element (|d| <= rtol |ref| + 1e-5 max |ref|, tests/helpers.py)"""
    assert_float_close(got, ref, rtol, what)


INT_KEYS = ["keys", "indexs", "sortBatchs", "cellIndexs", "startIndexs", "packedNeighs"]
EXACT_FLOAT_KEYS = ["aabbMin", "aabbMax", "sortPts", "sortFeatures"]  # pure copies / min / max


def compare_chain(g, o, pdf_rtol=RTOL):
    for k in INT_KEYS + EXACT_FLOAT_KEYS:
        assert g[k].shape == o[k].shape, (k, g[k].shape, o[k].shape)
        assert np.array_equal(g[k], o[k]), "%s differs (%d mismatches)" % (k, int((g[k] != o[k]).sum()))
    assert_close(g["pdfs"], o["pdfs"], pdf_rtol, "pdfs")
    for k in ("samplePts", "sampleBatchs", "sampleIndexs", "sampleFeatures", "transformedIndexs"):
        if k in o:
            assert g[k].shape == o[k].shape, (k, g[k].shape, o[k].shape)
            assert np.array_equal(g[k], o[k]), k


CASES = [
    # name, n_per, B, kind, ragged, radius, scaleInv, Fin, poisson_radius
    ("cfg0_uniform4096", 4096, 1, "uniform", False, 0.1, True, 3, 0.1),
    ("batched_sphere", 1024, 8, "sphere", False, 0.2, True, 1, 0.1),
    ("ragged_clustered", 700, 5, "clustered", True, 0.15, True, 4, 0.05),
    ("abs_radius_batched", 1500, 3, "uniform", True, 0.12, False, 3, 0.2),
    ("single_cell", 300, 4, "uniform", False, 1.2, True, 2, 1.3),
    ("tiny", 3, 2, "uniform", False, 0.5, True, 1, 0.5),
    # 5 x 80^3 = 2.56 M cells = 1250 scan tiles: past the single-pass (decoupled look-back) limit of 1024 tiles, the
    # cell-offset prefix sum takes the three-level form
    ("fine_grid_3level_scan", 1500, 5, "uniform", True, 0.0125, True, 1, 0.0125),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_grid_neighbors_pdf_poisson(mc, oracle, case):
    _, n_per, B, kind, ragged, radius, scaleInv, fin, prad = case
    pts, bids = make_cloud(n_per, B, 11, kind, ragged)
    feats = np.random.default_rng(3).random((len(pts), fin), dtype=np.float32)
    g = run_chain(mc, _wrap, _unwrap, pts, bids, feats, B, radius, scaleInv, poisson_radius=prad,
                  pdf_kwargs=dict(mode=0))
    o = run_chain(oracle, _ident, _ident, pts, bids, feats, B, radius, scaleInv, poisson_radius=prad)
    compare_chain(g, o, pdf_rtol=2e-6)  # mode 0 replays the reference arithmetic
    # the single-precision KDEs (mode 1: pair sums as Gram-matrix tiles on the matrix cores, the default; mode 2: the
    # subtract-first VALU loop) stay inside the feature-path tolerance -- per VALUE, not only against the largest one
    h = g["_handles"]
    for mode in (1, 2):
        fast = _unwrap(mc.compute_pdf(h["sP"], h["sB"], h["mn"], h["mx"], h["start"], h["packed"], 0.2, radius, B,
                                      scaleInv, mode=mode))
        assert_close(fast, o["pdfs"], RTOL, "pdfs(mode %d)" % mode)
        worst = float(np.max(np.abs(fast - o["pdfs"]) / np.abs(o["pdfs"]))) if fast.size else 0.0
        assert worst <= RTOL, (mode, worst)


def test_pdf_translated_cloud(mc, oracle):
    """Scene far from the origin (+500 m, absolute radius 0.1): the single-precision KDE subtracts raw coordinates
    before it scales (compute_pdf.cu:78-80), so its error does not grow with |p| / (R h)."""
    pts, bids = make_cloud(3000, 2, 17, "uniform")
    pts = (pts + np.float32(500.0)).astype(np.float32)
    feats = np.ones((len(pts), 1), np.float32)
    g = run_chain(mc, _wrap, _unwrap, pts, bids, feats, 2, 0.1, False, pdf_kwargs=dict(mode=0))
    o = run_chain(oracle, _ident, _ident, pts, bids, feats, 2, 0.1, False)
    compare_chain(g, o, pdf_rtol=2e-6)
    h = g["_handles"]
    for mode in (1, 2):  # mode 1 takes the row's first point as origin before it scales: nothing lost either
        fast = mc.compute_pdf(h["sP"], h["sB"], h["mn"], h["mx"], h["start"], h["packed"], 0.2, 0.1, 2, False, mode=mode)
        err = np.abs(_unwrap(fast) - o["pdfs"]).max() / np.abs(o["pdfs"]).max()
        assert err <= 2e-5, (mode, err)   # same error as at the origin (~1e-5), far inside RTOL


def test_dense_cells(mc, oracle):
    """A few very dense cells (>1000 points in one 27-window): exercises the streaming branch of the Poisson kernel,
    long cell segments in the stable ranking and long CSR rows."""
    rng = np.random.default_rng(13)
    blob = (0.5 + 0.012 * rng.normal(size=(900, 3))).astype(np.float32)
    rest = rng.random((400, 3), dtype=np.float32)
    pts = np.concatenate([blob, rest]).astype(np.float32)
    rng.shuffle(pts)
    bids = np.zeros((len(pts), 1), np.int32)
    feats = rng.random((len(pts), 2), dtype=np.float32)
    g = run_chain(mc, _wrap, _unwrap, pts, bids, feats, 1, 0.1, True, poisson_radius=0.1, pdf_kwargs=dict(mode=0))
    o = run_chain(oracle, _ident, _ident, pts, bids, feats, 1, 0.1, True, poisson_radius=0.1)
    klen = np.diff(np.append(o["startIndexs"][:, 0], len(o["packedNeighs"])))
    assert klen.max() > 800
    compare_chain(g, o, pdf_rtol=2e-6)
    # default KDE: rows of 65..192 points run several tiles per side, longer ones take the scalar loop
    assert ((klen > 64) & (klen <= 192)).any() and (klen > 192).any()
    _check_fast_pdf(mc, g["_handles"], o, 0.1, 1, True)


def _check_fast_pdf(mc, h, o, radius, B, scaleInv):
    for mode in (1, 2):
        fast = _unwrap(mc.compute_pdf(h["sP"], h["sB"], h["mn"], h["mx"], h["start"], h["packed"], 0.2, radius, B,
                                      scaleInv, mode=mode))
        worst = float(np.max(np.abs(fast - o["pdfs"]) / np.abs(o["pdfs"])))
        assert worst <= RTOL, (mode, worst)


def test_room_absolute_radius(mc, oracle):
    """Headline-like input: non-uniform room, absolute radius 0.1 (whole-batch box), 2 rooms x 20k points."""
    B = 2
    pts = np.concatenate([make_room(20000, 20180601), make_room(20000, 20180602)])
    bids = np.repeat(np.arange(B, dtype=np.int32), 20000).reshape(-1, 1)
    feats = np.random.default_rng(7).random((len(pts), 3), dtype=np.float32)
    g = run_chain(mc, _wrap, _unwrap, pts, bids, feats, B, 0.1, False, poisson_radius=0.2, pdf_kwargs=dict(mode=0))
    o = run_chain(oracle, _ident, _ident, pts, bids, feats, B, 0.1, False, poisson_radius=0.2)
    compare_chain(g, o, pdf_rtol=2e-6)
    _check_fast_pdf(mc, g["_handles"], o, 0.1, B, False)


def test_pooling_centres_differ_from_points(mc, oracle):
    """Pool-style search: centres are a different (smaller, unsorted) point set than the gridded points."""
    pts, bids = make_cloud(2000, 3, 5, "uniform")
    rng = np.random.default_rng(9)
    sel = np.sort(rng.choice(len(pts), 500, replace=False))
    centres = (pts[sel] + 0.01 * rng.normal(size=(500, 3))).astype(np.float32)  # may leave the box: clamped cells
    cb = bids[sel]
    feats = rng.random((len(pts), 2), dtype=np.float32)
    g = run_chain(mc, _wrap, _unwrap, pts, bids, feats, 3, 0.2, True, centres=centres, centre_bids=cb,
                  pdf_kwargs=dict(mode=0))
    o = run_chain(oracle, _ident, _ident, pts, bids, feats, 3, 0.2, True, centres=centres, centre_bids=cb)
    compare_chain(g, o, pdf_rtol=2e-6)


CONV_CASES = [
    # name, Fin, Fout, combin, avg, scaleInv, radius
    ("cfg0_3to8_combin", 3, 8, True, True, True, 0.1),
    ("1to16_combin", 1, 16, True, True, True, 0.15),
    ("dw32", 32, 32, False, True, True, 0.15),
    ("dw8_noavg_abs", 8, 8, False, False, False, 0.12),
    ("2to5_combin_padded", 2, 5, True, True, True, 0.15),  # 10 neurons -> nb=2, 6 padded neurons... 16 % 2 == 0
    ("4to6_combin_abs", 4, 6, True, True, False, 0.12),  # static (fin, fo) patterns: Fin = 2, 3, 4 each have their own
    ("3to16_combin", 3, 16, True, True, True, 0.15),  # 48 neurons -> nb = 6, r0 cycles 0, 2, 1, 0, 2, 1
    ("8to3_combin_generic", 8, 3, True, True, True, 0.15),  # Fin > 4: the generic gather path
    ("dw520_two_column_tiles", 520, 520, False, True, True, 0.3),  # nb = 65 > one launch's LDS weight tile: column tiles
    ("dw2048_column_tiles", 2048, 2048, False, True, True, 0.3),  # MCClassH Pool_3 at grow 64: nb = 256 (4 fwd / 6 bwd tiles)
    ("4to130_combin_valu_fallback", 4, 130, True, True, True, 0.3),  # combin, nb = 65 > MCCNN_LDS_MAX_NB: VALU kernels
    ("1to256_combin", 1, 256, True, True, True, 0.15),  # nb = 32 through the factored path
    ("1to13_combin_padded_noavg_abs", 1, 13, True, False, False, 0.12),  # factored Fin=1 path, 3 padded neurons
    ("1to64_combin_nostate", 1, 64, True, True, True, 0.15),  # backward without the forward's state: recomputed
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_spatial_conv_fwd_bwd(mc, oracle, case):
    import torch
    name, fin, fout, combin, avg, scaleInv, radius = case
    B = 2
    mc.KEEP_CONV_STATE = not name.endswith("nostate")
    pts, bids = make_cloud(300 if fin > 1024 else (1000 if fin > 256 else 1500), B, 21, "clustered", True)
    rng = np.random.default_rng(7)
    feats = (2 * rng.random((len(pts), fin)) - 1).astype(np.float32)
    o = run_chain(oracle, _ident, _ident, pts, bids, feats, B, radius, scaleInv, fout=fout, combin=combin)
    g = run_chain(mc, _wrap, _unwrap, pts, bids, feats, B, radius, scaleInv, fout=fout, combin=combin)
    for k in INT_KEYS:
        assert np.array_equal(g[k], o[k]), k
    w = o["mlp"]
    outF = fout if combin else fin
    og = (2 * np.random.default_rng(11).random((len(pts), outF)) - 1).astype(np.float32)
    # oracle
    args = (o["sortPts"], o["sortFeatures"], o["sortBatchs"], o["pdfs"], pts, o["startIndexs"], o["packedNeighs"],
            o["aabbMin"], o["aabbMax"], w["w1"], w["w2"], w["w3"], w["b1"], w["b2"], w["b3"])
    ref = oracle.spatial_conv(*args, fout, combin, B, radius, scaleInv, avg)
    rg = oracle.spatial_conv_grad(*args, og, fout, combin, B, radius, scaleInv, avg)
    # GPU through autograd (same pdfs as the oracle so that only the conv is compared)
    h = g["_handles"]
    tw = {k: _wrap(v).requires_grad_(True) for k, v in w.items()}
    sF = h["sF"].detach().clone().requires_grad_(True)
    out = mc.spatial_conv(h["sP"], sF, h["sB"], _wrap(o["pdfs"]), h["C"], h["start"], h["packed"], h["mn"], h["mx"],
                          tw["w1"], tw["w2"], tw["w3"], tw["b1"], tw["b2"], tw["b3"], fout, combin, B, radius,
                          scaleInv, avg)
    assert_close(_unwrap(out), ref, RTOL, "spatial_conv")
    out.backward(_wrap(og))
    torch.cuda.synchronize()
    mc.KEEP_CONV_STATE = True
    neurons = fin * fout if combin else fin
    got = [sF.grad, tw["w1"].grad, tw["b1"].grad, tw["w2"].grad, tw["b2"].grad, tw["w3"].grad, tw["b3"].grad]
    names = ["featGrad", "dw1", "db1", "dw2", "db2", "dw3", "db3"]
    # ReLU' = 1[pre >= 0] is discontinuous, so the pre-activations have to be bit-identical on both sides or a handful
    # of (edge, neuron) terms near zero take the other branch: the MFMA chains (zero start, bias as the last k-step,
    # correctly rounded delta = (p - c) / R) replay the oracle's fmaf chains exactly, and every gradient holds the
    # north-star tolerance.
    for nm, a, b in zip(names, got, rg):
        assert_close(_unwrap(a), b, RTOL, nm)
    # second implementation on the GPU: the VALU fallback kernels (every shape) and, for one input feature, the general
    # MFMA kernels instead of the factored ones -- selected per call through the library's test hook
    others = [(1, "VALU kernels")]
    if combin and fin == 1:
        others.append((2, "general MFMA kernels"))
    if not combin and fin % 8 == 0:
        others.append((4, "edge-streaming MFMA kernels"))   # the default for these layers is the row-per-lane form
    for mask, label in others:
        mc.debug_conv_impl(mask)
        try:
            tw2 = {k: _wrap(v).requires_grad_(True) for k, v in w.items()}
            sF2 = h["sF"].detach().clone().requires_grad_(True)
            out2 = mc.spatial_conv(h["sP"], sF2, h["sB"], _wrap(o["pdfs"]), h["C"], h["start"], h["packed"], h["mn"],
                                   h["mx"], tw2["w1"], tw2["w2"], tw2["w3"], tw2["b1"], tw2["b2"], tw2["b3"], fout,
                                   combin, B, radius, scaleInv, avg)
            out2.backward(_wrap(og))
            torch.cuda.synchronize()
        finally:
            mc.debug_conv_impl(0)
        assert_close(_unwrap(out), _unwrap(out2), 2e-5, label + ": forward")
        got2 = [sF2.grad, tw2["w1"].grad, tw2["b1"].grad, tw2["w2"].grad, tw2["b2"].grad, tw2["w3"].grad, tw2["b3"].grad]
        # all three GPU implementations share every pre-activation bit for bit; they differ in summation order only
        for nm, a, b in zip(names, got, got2):
            assert_close(_unwrap(a), _unwrap(b), 2e-5, label + ": " + nm)
    # padded output neurons: the library writes zeros (reference leaves them uninitialised)
    dw3 = _unwrap(tw["w3"].grad).reshape(-1)
    assert np.all(dw3[neurons * 8:] == 0)


@pytest.mark.parametrize("fin,fout,combin", [(1, 16, True), (3, 8, True), (16, 16, False)], ids=["f1", "combin", "dw"])
def test_conv_centres_without_neighbours(mc, oracle, fin, fout, combin):
    """Ragged neighbour lists: centres far from every point (empty rows at the start, in the middle and at the end
    of the list) give zero output rows and take no part in the gradients; one centre owns a very long row."""
    import torch
    B, radius = 1, 0.2
    pts, bids = make_cloud(1200, B, 33, "clustered", True)
    rng = np.random.default_rng(3)
    feats = (2 * rng.random((len(pts), fin)) - 1).astype(np.float32)
    far = np.array([[9.0, 9.0, 9.0]], np.float32)
    centres = np.concatenate([far, far + 1, pts[:300], far + 2, far + 3, far + 4, pts[300:500], far + 5]).astype(np.float32)
    cb = np.zeros((len(centres), 1), np.int32)
    o = run_chain(oracle, _ident, _ident, pts, bids, feats, B, radius, False, centres=centres, centre_bids=cb,
                  fout=fout, combin=combin)
    g = run_chain(mc, _wrap, _unwrap, pts, bids, feats, B, radius, False, centres=centres, centre_bids=cb, fout=fout,
                  combin=combin)
    for k in INT_KEYS:
        assert np.array_equal(g[k], o[k]), k
    deg = np.diff(np.append(o["startIndexs"].reshape(-1), len(o["packedNeighs"])))
    assert (deg == 0).sum() == 6 and deg.max() > 64
    w = o["mlp"]
    outF = fout if combin else fin
    og = (2 * np.random.default_rng(5).random((len(centres), outF)) - 1).astype(np.float32)
    args = (o["sortPts"], o["sortFeatures"], o["sortBatchs"], o["pdfs"], centres, o["startIndexs"], o["packedNeighs"],
            o["aabbMin"], o["aabbMax"], w["w1"], w["w2"], w["w3"], w["b1"], w["b2"], w["b3"])
    ref = oracle.spatial_conv(*args, fout, combin, B, radius, False, True)
    rg = oracle.spatial_conv_grad(*args, og, fout, combin, B, radius, False, True)
    h = g["_handles"]
    tw = {k: _wrap(v).requires_grad_(True) for k, v in w.items()}
    sF = h["sF"].detach().clone().requires_grad_(True)
    out = mc.spatial_conv(h["sP"], sF, h["sB"], _wrap(o["pdfs"]), h["C"], h["start"], h["packed"], h["mn"], h["mx"],
                          tw["w1"], tw["w2"], tw["w3"], tw["b1"], tw["b2"], tw["b3"], fout, combin, B, radius, False, True)
    got = _unwrap(out)
    assert np.all(got[deg == 0] == 0)
    assert_close(got, ref, RTOL, "spatial_conv")
    out.backward(_wrap(og))
    torch.cuda.synchronize()
    gg = [sF.grad, tw["w1"].grad, tw["b1"].grad, tw["w2"].grad, tw["b2"].grad, tw["w3"].grad, tw["b3"].grad]
    for nm, a, b in zip(["featGrad", "dw1", "db1", "dw2", "db2", "dw3", "db3"], gg, rg):
        assert_close(_unwrap(a), b, RTOL, nm)


@pytest.mark.parametrize("layout", ["slice_of_empties_large", "m_mod_64_is_1_small", "run_of_64_small"])
def test_depthwise_rows_with_whole_slices_of_empty_rows(mc, oracle, layout):
    """Row-per-lane depth-wise kernels: a slice (64 lanes) made ONLY of rows without a single edge has length 0 like the
    padding slices beyond the list -- its rows must still be stored as zeros (forward: centres without neighbours;
    backward: points that are nobody's neighbour). Empty rows sort to the end of their 1 024-row window, so such a slice
    appears whenever the window's non-empty count is a multiple of 64. The output buffers are taken from memory that
    held NaNs, so a row nobody writes shows."""
    import torch
    B, radius, fin = 1, 0.2, 16
    rng = np.random.default_rng(7)
    if layout == "slice_of_empties_large":
        # > 4 096 rows: the windowed layout. Window 0 = 960 centres with neighbours + 64 without
        n_near, n_far_c, n_c = 6000, 64, 5000
        near = rng.random((n_near, 3), dtype=np.float32)
        c_real = near[rng.permutation(n_near)[:n_c - n_far_c]]
        far_c = (5.0 + 3.0 * rng.random((n_far_c, 3))).astype(np.float32)
        centres = np.concatenate([c_real[:960], far_c, c_real[960:]]).astype(np.float32)
        lonely = (-5.0 - 3.0 * rng.random((128, 3))).astype(np.float32)   # points no centre reaches
        pts = np.concatenate([near[:3000], lonely, near[3000:]]).astype(np.float32)
    elif layout == "m_mod_64_is_1_small":
        near = rng.random((700, 3), dtype=np.float32)
        centres = np.concatenate([near[:192], np.array([[9.0, 9.0, 9.0]], np.float32)]).astype(np.float32)  # 193 rows, last empty
        # 64 lonely points: they sort to the FRONT of the grid order (lowest cells), slice 0 of the transposed plan is empty
        pts = np.concatenate([near, (-5.0 - rng.random((64, 3))).astype(np.float32)]).astype(np.float32)
    else:
        near = rng.random((900, 3), dtype=np.float32)
        far_c = (5.0 + 3.0 * rng.random((70, 3))).astype(np.float32)
        centres = np.concatenate([near[:128], far_c, near[128:300]]).astype(np.float32)   # 64-row slice [128, 192) all empty
        lonely = (-5.0 - 3.0 * rng.random((70, 3))).astype(np.float32)
        pts = np.concatenate([near[:256], lonely, near[256:]]).astype(np.float32)
    bids = np.zeros((len(pts), 1), np.int32)
    cb = np.zeros((len(centres), 1), np.int32)
    feats = (2 * rng.random((len(pts), fin)) - 1).astype(np.float32)
    o = run_chain(oracle, _ident, _ident, pts, bids, feats, B, radius, False, centres=centres, centre_bids=cb,
                  fout=fin, combin=False)
    g = run_chain(mc, _wrap, _unwrap, pts, bids, feats, B, radius, False, centres=centres, centre_bids=cb, fout=fin,
                  combin=False)
    for k in INT_KEYS:
        assert np.array_equal(g[k], o[k]), k
    deg = np.diff(np.append(o["startIndexs"].reshape(-1), len(o["packedNeighs"])))
    tdeg = np.bincount(o["packedNeighs"][:, 0], minlength=len(pts))
    assert (deg == 0).sum() >= 1 and (tdeg == 0).sum() >= 1
    w = o["mlp"]
    og = (2 * rng.random((len(centres), fin)) - 1).astype(np.float32)
    args = (o["sortPts"], o["sortFeatures"], o["sortBatchs"], o["pdfs"], centres, o["startIndexs"], o["packedNeighs"],
            o["aabbMin"], o["aabbMax"], w["w1"], w["w2"], w["w3"], w["b1"], w["b2"], w["b3"])
    ref = oracle.spatial_conv(*args, fin, False, B, radius, False, True)
    rg = oracle.spatial_conv_grad(*args, og, fin, False, B, radius, False, True)
    h = g["_handles"]
    assert mc._rows_shape(False, fin, h["sF"], len(centres), len(o["packedNeighs"]))  # the row kernels take this layer
    tw = {k: _wrap(v).requires_grad_(True) for k, v in w.items()}
    for rep in range(2):
        sF = h["sF"].detach().clone().requires_grad_(True)
        # poison the blocks the op's torch.empty() calls are about to receive
        junk = [torch.full((len(centres), fin), float("nan"), device="cuda"), torch.full((len(pts), fin), float("nan"), device="cuda")]
        torch.cuda.synchronize()
        del junk
        out = mc.spatial_conv(h["sP"], sF, h["sB"], _wrap(o["pdfs"]), h["C"], h["start"], h["packed"], h["mn"], h["mx"],
                              tw["w1"], tw["w2"], tw["w3"], tw["b1"], tw["b2"], tw["b3"], fin, False, B, radius, False, True)
        got = _unwrap(out)
        assert np.isfinite(got).all() and np.all(got[deg == 0] == 0)
        assert_close(got, ref, RTOL, "spatial_conv")
        junk = [torch.full((len(pts), fin), float("nan"), device="cuda"), torch.full((len(centres), fin), float("nan"), device="cuda")]
        torch.cuda.synchronize()
        del junk
        out.backward(_wrap(og))
        torch.cuda.synchronize()
        fg = _unwrap(sF.grad)
        assert np.isfinite(fg).all() and np.all(fg[tdeg == 0] == 0)
        assert_close(fg, rg[0], RTOL, "featGrad")
    for nm, a, b in zip(["dw1", "db1", "dw2", "db2", "dw3", "db3"],
                        [tw["w1"].grad, tw["b1"].grad, tw["w2"].grad, tw["b2"].grad, tw["w3"].grad, tw["b3"].grad], rg[1:]):
        assert_close(_unwrap(a) / 2.0, b, RTOL, nm)   # two backward passes accumulated


def test_poisson_dataflow_and_phased_forms_agree(mc, oracle):
    """All 27 colour phases in one launch (cells wait on per-cell flags) against one launch per phase and the oracle:
    uniform, clustered (cells with > 64 points, windows beyond the register path) and multi-cloud inputs."""
    for n, B, seed, kind, radius in ((4096, 1, 1, "uniform", 0.1), (3000, 3, 5, "clustered", 0.25), (600, 2, 9, "sphere", 0.6)):
        pts, bids = make_cloud(n, B, seed, kind, True)
        feats = np.ones((len(pts), 1), np.float32)
        o = run_chain(oracle, _ident, _ident, pts, bids, feats, B, radius, True, poisson_radius=radius)
        outs = []
        for flag in (True, False):
            mc.POISSON_DATAFLOW = flag
            try:
                g = run_chain(mc, _wrap, _unwrap, pts, bids, feats, B, radius, True, poisson_radius=radius)
            finally:
                mc.POISSON_DATAFLOW = True
            outs.append(g)
            for k in ("samplePts", "sampleBatchs", "sampleIndexs"):
                assert np.array_equal(g[k], o[k]), (kind, flag, k)
        for k in ("samplePts", "sampleBatchs", "sampleIndexs"):
            assert np.array_equal(outs[0][k], outs[1][k])


def test_poisson_dataflow_stress_on_a_large_fine_grid(mc, oracle):
    """The single-launch sampling hands selection bytes between cells of different XCDs through relaxed agent-scope
    atomics ordered by s_waitcnt (no L2-wide fences). Stress: BASELINE cfg3's finest level -- 131 072
    points in 16 x 40^3 = 1.02 M cells, thousands of dependent cells in flight -- ten times in a row beside other work on
    the GPU; every run must give the phased form's (and the oracle's) sample set and order."""
    import torch
    from mccnn_amd.workloads import modelnet_like
    pts, bids = modelnet_like(8192, 16, 47)
    B, radius = 16, 0.025
    P, Bi = _wrap(pts), _wrap(bids)
    mn, mx = mc.compute_aabb(P, Bi, B, True)
    keys, idx = mc.sort_points_step1(P, Bi, mn, mx, B, radius, True)
    feats = torch.zeros((len(pts), 1), device="cuda")
    sP, sB, _, cells = mc.sort_points_step2(P, Bi, feats, keys, idx, mn, mx, B, radius, True)
    omn, omx = oracle.compute_aabb(pts, bids, B, True)
    ok, oi = oracle.sort_points_step1(pts, bids, omn, omx, B, radius, True)
    osp, osb, _, ocl = oracle.sort_points_step2(pts, bids, np.zeros((len(pts), 1), np.float32), ok, oi, omn, omx, B, radius, True)
    rp, rb, ri = oracle.poisson_sampling(osp, osb, ocl, omn, omx, radius, B, True)
    mc.POISSON_DATAFLOW = False
    try:
        pp, pb, pi = mc.poisson_sampling(sP, sB, cells, mn, mx, radius, B, True)
    finally:
        mc.POISSON_DATAFLOW = True
    assert np.array_equal(_unwrap(pi), ri) and np.array_equal(_unwrap(pp), rp)
    before = mc.POISSON_FALLBACKS
    noise = torch.rand((4096, 4096), device="cuda")
    for rep in range(10):
        noise = noise @ noise.t() * 1e-4      # uneven load beside the sampling
        dp, db, di = mc.poisson_sampling(sP, sB, cells, mn, mx, radius, B, True)
        assert torch.equal(di, pi) and torch.equal(dp, pp) and torch.equal(db, pb), rep
    assert mc.POISSON_FALLBACKS == before      # ... and by the single-launch form, not its fallback
    torch.cuda.synchronize()


def test_poisson_timeout_falls_back_to_phased_form(mc, oracle):
    """The dataflow kernel relies on earlier-phase cells finishing while later ones wait (bounded spin). Mode 2 removes
    the spin altogether: a cell whose predecessor is not done yet raises the failure flag, the count reports -1 and the
    op repeats with one launch per phase -- same samples, same order."""
    pts, bids = make_cloud(4096, 2, 3, "uniform", True)
    feats = np.ones((len(pts), 1), np.float32)
    o = run_chain(oracle, _ident, _ident, pts, bids, feats, 2, 0.1, True, poisson_radius=0.1)
    before = mc.POISSON_FALLBACKS
    mc.POISSON_DATAFLOW = 2
    try:
        g = run_chain(mc, _wrap, _unwrap, pts, bids, feats, 2, 0.1, True, poisson_radius=0.1)
    finally:
        mc.POISSON_DATAFLOW = True
    assert mc.POISSON_FALLBACKS == before + 1          # the fallback really ran
    for k in ("samplePts", "sampleBatchs", "sampleIndexs", "sampleFeatures", "transformedIndexs"):
        assert np.array_equal(g[k], o[k]), k


def test_batch_id_validation(mc):
    """Batch ids outside [0, batchSize): counted by mccnn_check_batch_ids, raised as MCCNN_E_BATCHID by the binding when
    CHECK_BATCH_IDS is on; the kernels clamp ids, so the chain stays memory-safe either way."""
    import torch
    from mccnn_amd._lib import MCCNNError
    pts, bids = make_cloud(256, 2, 0)
    bad = bids.copy()
    bad[5, 0] = 7
    bad[9, 0] = -1
    P, Bi = _wrap(pts), _wrap(bad)
    assert mc.check_batch_ids(_wrap(bids), 2) == 0 and mc.check_batch_ids(Bi, 2) == 2
    mc.CHECK_BATCH_IDS = True
    try:
        with pytest.raises(MCCNNError, match="batch id"):
            mc.compute_aabb(P, Bi, 2, True)
    finally:
        mc.CHECK_BATCH_IDS = False
    mn, mx = mc.compute_aabb(P, Bi, 2, True)                       # invalid ids are skipped by the box ...
    k, i = mc.sort_points_step1(P, Bi, mn, mx, 2, 0.2, True)       # ... and clamped by the grid: still a permutation
    torch.cuda.synchronize()
    assert np.array_equal(np.sort(_unwrap(i)), np.arange(len(pts)))


def test_permutation_ops_and_adjoints(mc, oracle):
    import torch
    rng = np.random.default_rng(2)
    n, F = 1000, 5
    perm = rng.permutation(n).astype(np.int32)
    f = rng.random((n, F), dtype=np.float32)
    assert np.array_equal(_unwrap(mc.sort_features(_wrap(f), _wrap(perm))), oracle.sort_features(f, perm))
    assert np.array_equal(_unwrap(mc.sort_features_back(_wrap(f), _wrap(perm))), oracle.sort_features_back(f, perm))
    # mutual adjoints (MCConvModuleSrc:37-45)
    x = _wrap(f).requires_grad_(True)
    y = mc.sort_features(x, _wrap(perm))
    gy = rng.random((n, F), dtype=np.float32)
    y.backward(_wrap(gy))
    assert np.array_equal(_unwrap(x.grad), oracle.sort_features_back(gy, perm))
    # sampled features + scatter-with-zero-fill gradient (MCConvModuleSrc:63-68)
    idx = np.sort(rng.choice(n, 100, replace=False)).astype(np.int32)
    x2 = _wrap(f).requires_grad_(True)
    s = mc.get_sampled_features(_wrap(idx), x2)
    assert np.array_equal(_unwrap(s), oracle.get_sampled_features(idx, f))
    gs = rng.random((100, F), dtype=np.float32)
    s.backward(_wrap(gs))
    assert np.array_equal(_unwrap(x2.grad), oracle.get_sampled_features_grad(idx, f, gs))
    assert np.array_equal(_unwrap(mc.transform_indexs(_wrap(idx), _wrap(perm))), oracle.transform_indexs(idx, perm))


def test_sort_step2_gradient_routing(mc, oracle):
    pts, bids = make_cloud(500, 2, 4, "uniform")
    feats = np.random.default_rng(1).random((len(pts), 3), dtype=np.float32)
    P, Bi = _wrap(pts).requires_grad_(True), _wrap(bids)
    F = _wrap(feats).requires_grad_(True)
    mn, mx = mc.compute_aabb(P, Bi, 2, True)
    k, i = mc.sort_points_step1(P, Bi, mn, mx, 2, 0.2, True)
    sP, sB, sF, cells = mc.sort_points_step2(P, Bi, F, k, i, mn, mx, 2, 0.2, True)
    gp = np.random.default_rng(5).random(pts.shape, dtype=np.float32)
    gf = np.random.default_rng(6).random(feats.shape, dtype=np.float32)
    (sP * _wrap(gp)).sum().backward(retain_graph=True)
    (sF * _wrap(gf)).sum().backward()
    idx = _unwrap(i)
    rp, rf = oracle.sort_points_step2_grad(idx, gp, gf)
    assert np.array_equal(_unwrap(P.grad), rp) and np.array_equal(_unwrap(F.grad), rf)


def test_invalid_arguments_raise(mc):
    import torch
    pts, bids = make_cloud(64, 1, 0)
    P, Bi = _wrap(pts), _wrap(bids)
    with pytest.raises(mc.InvalidArgumentError):
        mc.compute_aabb(P[:, :2].contiguous(), Bi, 1, True)  # not 3 components (aabb_gpu.cc:53-58)
    with pytest.raises(mc.InvalidArgumentError):
        mc.compute_aabb(P, Bi.reshape(-1), 1, True)          # batch ids must be [N,1]
    mn, mx = mc.compute_aabb(P, Bi, 1, True)
    with pytest.raises(mc.InvalidArgumentError):
        mc.find_neighbors(P, Bi, P, torch.zeros((1, 2, 2, 2, 2), dtype=torch.int32).cuda(), mn, mx, -1.0, 1, True)


def test_empty_inputs(mc):
    import torch
    P = torch.zeros((0, 3), dtype=torch.float32).cuda()
    Bi = torch.zeros((0, 1), dtype=torch.int32).cuda()
    mn, mx = mc.compute_aabb(P, Bi, 2, True)
    assert np.all(_unwrap(mn) == np.finfo(np.float32).max) and np.all(_unwrap(mx) == -np.finfo(np.float32).max)


def test_pdf_row_lengths_at_tile_and_plane_boundaries(mc, oracle):
    """KDE rows of exactly k points for k around the 16-point tile edge, the 64-point gather chunk and the 192-point LDS
    plane capacity of the matrix-core kernel (rows above it take its subtract-first loop): clusters of k points, each
    inside a ball of diameter < r and far from the next one, so that every centre of a cluster has the whole cluster --
    and nothing else -- as its row."""
    # (257 .. 700: long rows on the workgroup's POOLED planes, 16 x 192 points here; rows beyond a pool take the streamed
    # loop -- covered by the next test, where the pool is 768 points)
    ks = [1, 2, 3, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 80, 100, 127, 128, 129, 191, 192, 193, 257, 400, 700]
    rng = np.random.default_rng(23)
    r = 0.1
    pts = []
    for n, k in enumerate(ks):
        centre = np.array([0.5 * (n % 5), 0.5 * ((n // 5) % 5), 0.5 * (n // 25)]) + 0.05
        d = rng.normal(size=(k, 3))
        d *= (0.045 * rng.random((k, 1)) ** (1 / 3)) / np.linalg.norm(d, axis=1, keepdims=True)
        pts.append(centre + d)
    pts = np.concatenate(pts).astype(np.float32)
    perm = rng.permutation(len(pts))
    pts = pts[perm]
    bids = np.zeros((len(pts), 1), np.int32)
    feats = np.ones((len(pts), 1), np.float32)
    g = run_chain(mc, _wrap, _unwrap, pts, bids, feats, 1, r, False, pdf_kwargs=dict(mode=0))
    o = run_chain(oracle, _ident, _ident, pts, bids, feats, 1, r, False)
    klen = np.diff(np.append(o["startIndexs"][:, 0], len(o["packedNeighs"])))
    assert sorted(set(klen.tolist())) == sorted(ks)
    compare_chain(g, o, pdf_rtol=2e-6)
    _check_fast_pdf(mc, g["_handles"], o, r, 1, False)


def test_pdf_long_rows_in_a_list_of_many_rows(mc, oracle):
    """Lists of >= 16 384 rows take the four-wave form of the KDE kernel (four rows per wave, pooled planes of 768 points):
    clusters of 300 / 700 (pooled) and 900 (streamed) points among 17 576 isolated ones."""
    rng = np.random.default_rng(29)
    r = 0.1
    g1 = np.arange(26, dtype=np.float64) * 0.3
    lattice = np.stack(np.meshgrid(g1, g1, g1, indexing="ij"), -1).reshape(-1, 3) + 0.02 * rng.random((26 ** 3, 3))
    pts = [lattice]
    ks = [300, 700, 900]
    for n, k in enumerate(ks):
        centre = np.array([0.15 + 0.3 * (3 + 5 * n), 0.15 + 0.3 * 7, 0.15 + 0.3 * 11])  # the middle of a lattice cube
        d = rng.normal(size=(k, 3))
        d *= (0.045 * rng.random((k, 1)) ** (1 / 3)) / np.linalg.norm(d, axis=1, keepdims=True)
        pts.append(centre + d)
    pts = np.concatenate(pts).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    bids = np.zeros((len(pts), 1), np.int32)
    feats = np.ones((len(pts), 1), np.float32)
    g = run_chain(mc, _wrap, _unwrap, pts, bids, feats, 1, r, False, pdf_kwargs=dict(mode=0))
    o = run_chain(oracle, _ident, _ident, pts, bids, feats, 1, r, False)
    klen = np.diff(np.append(o["startIndexs"][:, 0], len(o["packedNeighs"])))
    assert sorted(set(klen.tolist())) == [1] + ks
    compare_chain(g, o, pdf_rtol=2e-6)
    _check_fast_pdf(mc, g["_handles"], o, r, 1, False)
