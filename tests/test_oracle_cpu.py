"""CPU tests of the oracle (test infrastructure): known answers that DO exist in the reference, an independent
NumPy float64 cross-check, finite-difference gradients, invariants, and the committed golden fixtures."""
import glob
import os

import numpy as np
import pytest

from tests.helpers import make_cloud, make_mlp, conv_nb, run_chain

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ident = lambda a: a


# ------------------------------------------------------------------ structural known answers
# The expected values are NOT typed in here: tests/golden/reference_constants.json is written by
# tests/golden/make_reference_constants.py, which parses them out of the reference's own sources in the build container
# (find_neighbors.cu:282-291, poisson_sampling.cu:192-196, compute_pdf.cu:85-87, genCompileScript.py:20,
# sort_gpu.cu:404-408 + the radii of models/*.py).
import json

with open(os.path.join(GOLD, "reference_constants.json")) as _f:
    REFC = json.load(_f)


def test_neighbor_offset_table(oracle):
    t = oracle.cell_offsets()
    assert t.tolist() == REFC["cell_offsets"]
    assert len({tuple(r) for r in t.tolist()}) == 27 and np.abs(t).max() == 1
    # the closed form the HIP kernels use (common.h neigh_offset): x fastest (+1, 0, -1), then y, then z
    exp = np.array([[1 - o % 3, 1 - (o // 3) % 3, 1 - o // 9] for o in range(27)])
    assert np.array_equal(np.array(REFC["cell_offsets"]), exp)


def test_poisson_phase_table(oracle):
    t = oracle.cell_offsets_pool()
    assert t.tolist() == REFC["cell_offsets_pool"]
    assert len({tuple(r) for r in t.tolist()}) == 27 and np.abs(t).max() == 1


def test_poisson_phase_table_of_the_kernels():
    # the table compiled into the HIP kernels (csrc/common.h kPoolOffsets, packed (dx+1) | (dy+1)<<2 | (dz+1)<<4)
    import re
    src = open(os.path.join(os.path.dirname(GOLD), "..", "mccnn_amd", "csrc", "common.h")).read()
    body = src[src.index("kPoolOffsets[27]"):]
    body = body[:body.index("};")]
    ent = re.findall(r"(\d)\s*\|\s*\((\d)\s*<<\s*2\)\s*\|\s*\((\d)\s*<<\s*4\)", body)
    assert len(ent) == 27
    assert [[int(a) - 1, int(b) - 1, int(c) - 1] for a, b, c in ent] == REFC["cell_offsets_pool"]


@pytest.mark.parametrize("r,nc", [tuple(x) for x in REFC["num_cells_known"]])
def test_num_cells_known_answers(oracle, r, nc):
    # numCells(scale_inv) = max(1, (int)(1.0f / r)), sort_gpu.cu:404-408, on every radius the reference's models use
    z = np.zeros((1, 3), np.float32)
    assert oracle.num_cells(z, z + 1, 1, r, True) == nc


def test_num_cells_absolute(oracle):
    mn = np.array([[0, 0, 0]], np.float32)
    mx = np.array([[6.0, 4.0, 2.8]], np.float32)
    assert oracle.num_cells(mn, mx, 1, 0.1, False) == int(np.float32(6.0) / np.float32(0.1))
    assert oracle.num_cells(mn, mx, 1, 100.0, False) == 1  # 0 -> 1


def test_block_size(oracle):
    assert oracle.get_block_size() == REFC["block_mlp_size"]  # genCompileScript.py:20


def test_gauss_constant(oracle):
    # compute_pdf.cu:85-87: a centre whose only neighbour is itself has pdf = (invH * c * exp(0))^3 / 1, every product
    # rounded to f32 as the reference does; c is the literal parsed from the reference
    c = float(REFC["gauss_norm"])
    assert REFC["gauss_norm_uses"] == 3
    pts = np.array([[0.5, 0.5, 0.5]], np.float32)
    bids = np.zeros((1, 1), np.int32)
    mn, mx = pts.copy(), pts.copy() + 1
    start = np.zeros((1, 1), np.int32)
    packed = np.zeros((1, 2), np.int32)
    for window in (0.2, 0.25):
        pdf = oracle.compute_pdf(pts, bids, mn, mx, start, packed, window, 0.3, 1, False)
        invH = np.float32(1) / np.float32(window)          # float invH = 1 / h
        g = np.float32(float(invH) * (c * 1.0))            # float * (double * exp(0)) -> double, rounded on assignment
        g = np.float32(float(g * invH) * (c * 1.0))        # (float * float) is a float product in C, then * double
        g = np.float32(float(g * invH) * (c * 1.0))
        assert np.asarray(pdf).reshape(-1)[0] == g


# ------------------------------------------------------------------ NumPy float64 cross-check
def _brute_neighbors(centres, cb, pts, pb, R_of_b):
    out = []
    for i in range(len(centres)):
        b = cb[i, 0]
        d = np.linalg.norm(pts.astype(np.float64) - centres[i].astype(np.float64), axis=1)
        R = R_of_b[b]
        same = pb[:, 0] == b
        sure = np.nonzero(same & (d < R * (1 - 1e-6)))[0]
        maybe = np.nonzero(same & (np.abs(d - R) <= R * 1e-6))[0]
        out.append((set(sure.tolist()), set(maybe.tolist())))
    return out


@pytest.mark.parametrize("scaleInv,radius", [(True, 0.15), (False, 0.12)])
def test_chain_against_numpy(oracle, scaleInv, radius):
    B = 3
    pts, bids = make_cloud(400, B, 3, "clustered", True)
    fin, fout = 2, 3
    feats = (2 * np.random.default_rng(1).random((len(pts), fin)) - 1).astype(np.float32)
    o = run_chain(oracle, ident, ident, pts, bids, feats, B, radius, scaleInv, fout=fout, combin=True)
    # aabb
    for b in range(B):
        sel = pts[bids[:, 0] == b]
        if scaleInv:
            assert np.array_equal(o["aabbMin"][b], sel.min(0)) and np.array_equal(o["aabbMax"][b], sel.max(0))
        else:
            assert np.array_equal(o["aabbMin"][b], pts.min(0)) and np.array_equal(o["aabbMax"][b], pts.max(0))
    ext = (o["aabbMax"] - o["aabbMin"]).max(1)
    R_of_b = (radius * ext if scaleInv else np.full(B, radius)).astype(np.float64)
    # sort: permutation, stable inside a cell, keys monotone, cell table partitions [0, N)
    idx, keys = o["indexs"], o["keys"]
    assert sorted(idx.tolist()) == list(range(len(pts)))
    skeys = np.empty_like(keys)
    skeys[idx] = keys
    assert np.all(np.diff(skeys) >= 0)
    inv = np.argsort(idx)
    for k in np.unique(keys):
        members = inv[skeys == k]
        assert np.all(np.diff(members) > 0)  # ascending original index inside a cell
    cells = o["cellIndexs"].reshape(-1, 2)
    ne = cells[cells[:, 1] > cells[:, 0]]
    assert (ne[:, 1] - ne[:, 0]).sum() == len(pts) and np.array_equal(np.sort(ne[:, 0])[1:], np.sort(ne[:, 1])[:-1])
    assert np.array_equal(o["sortPts"], pts[inv]) and np.array_equal(o["sortBatchs"], bids[inv])
    # neighbours vs brute force (pairs within 1e-6 R of the sphere are exempt)
    start, packed = o["startIndexs"][:, 0], o["packedNeighs"]
    sb = o["sortBatchs"]
    brute = _brute_neighbors(pts, bids, o["sortPts"], sb, R_of_b)
    assert np.array_equal(np.unique(packed[:, 1]), np.unique(packed[:, 1])) and np.all(np.diff(packed[:, 1]) >= 0)
    for i in range(len(pts)):
        e0 = start[i]
        e1 = start[i + 1] if i + 1 < len(pts) else len(packed)
        got = set(packed[e0:e1, 0].tolist())
        sure, maybe = brute[i]
        assert sure <= got <= (sure | maybe), i
        assert np.all(packed[e0:e1, 1] == i)
    # pdf: closed form in float64
    sp = o["sortPts"].astype(np.float64)
    h = 0.2
    pdf_ref = np.empty(len(packed))
    for i in range(len(pts)):
        e0 = start[i]
        e1 = start[i + 1] if i + 1 < len(pts) else len(packed)
        nb = packed[e0:e1, 0]
        if len(nb) == 0:
            continue
        R = R_of_b[bids[i, 0]]
        d = (sp[nb][None, :, :] - sp[nb][:, None, :]) / (R * h)
        g = np.prod((1 / h) * 0.39894228 * np.exp(-0.5 * d * d), axis=2)
        pdf_ref[e0:e1] = g.sum(1) / len(nb)
    assert np.allclose(o["pdfs"][:, 0], pdf_ref, rtol=2e-5, atol=0)
    # conv forward / backward against a float64 NumPy implementation + finite differences
    w = o["mlp"]
    og = (2 * np.random.default_rng(4).random((len(pts), fout)) - 1).astype(np.float32)
    args = (o["sortPts"], o["sortFeatures"], o["sortBatchs"], o["pdfs"], pts, o["startIndexs"], packed, o["aabbMin"],
            o["aabbMax"], w["w1"], w["w2"], w["w3"], w["b1"], w["b2"], w["b3"])
    out = oracle.spatial_conv(*args, fout, True, B, radius, scaleInv, True)
    grads = oracle.spatial_conv_grad(*args, og, fout, True, B, radius, scaleInv, True)

    def np_conv(w1, b1, w2, b2, w3, b3, F):
        nb = conv_nb(fin, fout, True)
        j, i = packed[:, 0], packed[:, 1]
        K = np.diff(np.append(start, len(packed))).astype(np.float64)
        delta = (sp[j] - pts[i].astype(np.float64)) / R_of_b[sb[j, 0]][:, None]
        c = o["pdfs"][:, 0].astype(np.float64) * K[i]
        res = np.zeros((len(pts), fout))
        W1 = w1.reshape(-1).reshape(8 * nb, 3)
        for q in range(nb):
            h1 = np.maximum(delta @ W1[8 * q:8 * q + 8].T + b1[8 * q:8 * q + 8], 0)
            W2 = w2.reshape(-1)[64 * q:64 * q + 64].reshape(8, 8)
            h2 = np.maximum(h1 @ W2.T + b2[8 * q:8 * q + 8], 0)
            W3 = w3.reshape(-1)[64 * q:64 * q + 64].reshape(8, 8)
            ov = h2 @ W3.T + b3[8 * q:8 * q + 8]
            for n in range(8):
                nu = 8 * q + n
                if nu < fin * fout:
                    np.add.at(res, (i, nu // fin), F[j, nu % fin] * ov[:, n] / c)
        return res

    W = {k: v.astype(np.float64) for k, v in w.items()}
    F64 = o["sortFeatures"].astype(np.float64)
    ref = np_conv(W["w1"], W["b1"], W["w2"], W["b2"], W["w3"], W["b3"], F64)
    assert np.abs(out - ref).max() <= 1e-5 * np.abs(ref).max()

    # backward, independent vectorised float64 back-propagation (ReLU' = 1[pre >= 0], spatial_conv.cu:404,429)
    nb = conv_nb(fin, fout, True)
    j, i = packed[:, 0], packed[:, 1]
    K = np.diff(np.append(start, len(packed))).astype(np.float64)
    delta = (sp[j] - pts[i].astype(np.float64)) / R_of_b[sb[j, 0]][:, None]
    c = o["pdfs"][:, 0].astype(np.float64) * K[i]
    ref_g = dict(F=np.zeros_like(F64), w1=np.zeros(24 * nb), b1=np.zeros(8 * nb), w2=np.zeros(64 * nb),
                 b2=np.zeros(8 * nb), w3=np.zeros(64 * nb), b3=np.zeros(8 * nb))
    for q in range(nb):
        W1 = W["w1"].reshape(-1)[24 * q:24 * q + 24].reshape(8, 3)
        W2 = W["w2"].reshape(-1)[64 * q:64 * q + 64].reshape(8, 8)
        W3 = W["w3"].reshape(-1)[64 * q:64 * q + 64].reshape(8, 8)
        pre1 = delta @ W1.T + W["b1"][8 * q:8 * q + 8]
        a1 = np.maximum(pre1, 0)
        pre2 = a1 @ W2.T + W["b2"][8 * q:8 * q + 8]
        a2 = np.maximum(pre2, 0)
        ov = a2 @ W3.T + W["b3"][8 * q:8 * q + 8]
        do = np.zeros((len(packed), 8))
        for n in range(8):
            nu = 8 * q + n
            if nu < fin * fout:
                do[:, n] = og[i, nu // fin] * F64[j, nu % fin] / c
                np.add.at(ref_g["F"], (j, nu % fin), og[i, nu // fin] * ov[:, n] / c)
        ref_g["w3"][64 * q:64 * q + 64] = (do.T @ a2).reshape(-1)
        ref_g["b3"][8 * q:8 * q + 8] = do.sum(0)
        dpre2 = (do @ W3) * (pre2 >= 0)
        ref_g["w2"][64 * q:64 * q + 64] = (dpre2.T @ a1).reshape(-1)
        ref_g["b2"][8 * q:8 * q + 8] = dpre2.sum(0)
        dpre1 = (dpre2 @ W2) * (pre1 >= 0)
        ref_g["w1"][24 * q:24 * q + 24] = (dpre1.T @ delta).reshape(-1)
        ref_g["b1"][8 * q:8 * q + 8] = dpre1.sum(0)
    for nm, g in zip(["F", "w1", "b1", "w2", "b2", "w3", "b3"], grads):
        r = ref_g[nm].reshape(-1)
        assert np.abs(g.reshape(-1) - r).max() <= 2e-5 * max(np.abs(r).max(), 1e-6), nm
    # the output is LINEAR in the features and in (w3, b3): central differences are exact there (no ReLU kink)
    loss = lambda **kw: float((np_conv(**{**dict(w1=W["w1"], b1=W["b1"], w2=W["w2"], b2=W["b2"], w3=W["w3"], b3=W["b3"], F=F64), **kw}) * og).sum())
    rng = np.random.default_rng(8)
    base = dict(F=F64, **W)
    for nm, g in (("F", grads[0]), ("w3", grads[5]), ("b3", grads[6])):
        flat = g.reshape(-1)
        for _ in range(4):
            k = int(rng.integers(0, flat.size))
            if nm in ("w3", "b3") and ((k // 8 if nm == "w3" else k) >= fin * fout):
                continue  # padded output neuron: no gradient
            eps = 1e-3
            hi, lo = base[nm].copy().reshape(-1), base[nm].copy().reshape(-1)
            hi[k] += eps
            lo[k] -= eps
            fd = (loss(**{nm: hi.reshape(base[nm].shape)}) - loss(**{nm: lo.reshape(base[nm].shape)})) / (2 * eps)
            assert abs(fd - flat[k]) <= 1e-4 * max(np.abs(flat).max(), 1e-3), (nm, k, fd, flat[k])


# ------------------------------------------------------------------ Poisson sampling: properties + re-simulation
def test_poisson_properties_and_resimulation(oracle):
    B, radius = 2, 0.2
    pts, bids = make_cloud(250, B, 9, "uniform")
    mn, mx = oracle.compute_aabb(pts, bids, B, True)
    k, idx = oracle.sort_points_step1(pts, bids, mn, mx, B, radius, True)
    sp, sb, _, cells = oracle.sort_points_step2(pts, bids, pts, k, idx, mn, mx, B, radius, True)
    s_pts, s_b, s_idx = oracle.poisson_sampling(sp, sb, cells, mn, mx, radius, B, True)
    ext = (mx - mn).max(1)
    assert np.array_equal(s_pts, sp[s_idx]) and np.array_equal(s_b, sb[s_idx]) and len(set(s_idx.tolist())) == len(s_idx)
    for b in range(B):
        R = np.float32(radius) * ext[b]
        sel = s_pts[s_b[:, 0] == b].astype(np.float64)
        d = np.linalg.norm(sel[:, None] - sel[None], axis=2) + np.eye(len(sel)) * 10
        assert d.min() >= R * (1 - 1e-6)                      # separated
        rest = sp[(sb[:, 0] == b)].astype(np.float64)
        dm = np.linalg.norm(rest[:, None] - sel[None], axis=2).min(1)
        assert dm.max() < R * (1 + 1e-6)                      # maximal: every point is covered
    # independent pure-Python greedy in the canonical order (batch -> phase -> launch-linear cell -> point)
    pool = oracle.cell_offsets_pool()
    nc = cells.shape[1]
    G = -(-nc // 3)
    nB = -(-G // 4)
    chosen, flag = [], np.zeros(len(sp), bool)
    for b in range(B):
        R = np.float32(np.float32(radius) * ext[b])
        for ph in range(27):
            for lin in range((4 * nB) ** 3):
                blk, thr = divmod(lin, 64)
                bx, by, bz = blk % nB, (blk // nB) % nB, blk // (nB * nB)
                tx, ty, tz = thr % 4, (thr // 4) % 4, thr // 16
                c = [3 * (t + 4 * bb) + 1 + int(o) for t, bb, o in zip((tx, ty, tz), (bx, by, bz), pool[ph])]
                if max(c) >= nc:
                    continue
                p0, p1 = cells[b, c[0], c[1], c[2]]
                for i in range(p0, p1):
                    ok = True
                    for o in pool:
                        cc = [c[0] + int(o[0]), c[1] + int(o[1]), c[2] + int(o[2])]
                        if min(cc) < 0 or max(cc) >= nc:
                            continue
                        j0, j1 = cells[b, cc[0], cc[1], cc[2]]
                        for j in range(j0, j1):
                            if flag[j]:
                                dv = sp[j] - sp[i]
                                if np.sqrt(np.float32(dv[0] * dv[0] + dv[1] * dv[1]) + dv[2] * dv[2]) < R:
                                    ok = False
                    if ok:
                        flag[i] = True
                        chosen.append(i)
    assert chosen == s_idx.tolist()


# ------------------------------------------------------------------ golden fixtures (regression pins)
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "chain_*.npz"))), ids=os.path.basename)
def test_oracle_matches_golden(oracle, path):
    g = np.load(path)
    B, radius, scaleInv, fin, fout, combin, prad = g["attrs"]
    B, scaleInv, fin, fout, combin = int(B), bool(scaleInv), int(fin), int(fout), bool(combin)
    o = run_chain(oracle, ident, ident, g["in_points"], g["in_batch_ids"], g["in_features"], B, float(radius), scaleInv,
                  fout=fout, combin=combin, poisson_radius=float(prad))
    for k in ("keys", "indexs", "cellIndexs", "startIndexs", "packedNeighs", "sampleIndexs", "transformedIndexs",
              "sampleBatchs", "sortBatchs"):
        assert np.array_equal(o[k], g[k]), k
    assert np.allclose(o["pdfs"], g["pdfs"], rtol=1e-6)
    w = {k: g["mlp_" + k] for k in ("w1", "b1", "w2", "b2", "w3", "b3")}
    args = (o["sortPts"], o["sortFeatures"], o["sortBatchs"], o["pdfs"], g["in_points"], o["startIndexs"],
            o["packedNeighs"], o["aabbMin"], o["aabbMax"], w["w1"], w["w2"], w["w3"], w["b1"], w["b2"], w["b3"])
    out = oracle.spatial_conv(*args, fout, combin, B, float(radius), scaleInv, True)
    assert np.allclose(out, g["conv_out"], rtol=1e-5, atol=1e-6)


def test_reference_generated_cloud_fixture():
    p = np.load(os.path.join(GOLD, "nonuniform_cloud.npz"))["points"]
    assert p.shape == (4096, 3) and p.dtype == np.float32
    # gradient protocol (utils/DataSet.py:431-492): density grows along the longest axis
    x = p[:, np.argmax(p.max(0) - p.min(0))]
    lo, hi = np.percentile(x, [0, 100])
    first, last = np.sum(x < lo + 0.25 * (hi - lo)), np.sum(x > hi - 0.25 * (hi - lo))
    assert last > 3 * first


def test_openmp_build_gives_identical_integers():
    from oracle.oracle import Oracle
    seq, omp = Oracle(), Oracle(omp=True)
    pts, bids = make_cloud(500, 2, 2, "uniform")
    feats = pts.copy()
    a = run_chain(seq, ident, ident, pts, bids, feats, 2, 0.2, True)
    b = run_chain(omp, ident, ident, pts, bids, feats, 2, 0.2, True)
    for k in ("keys", "indexs", "cellIndexs", "startIndexs", "packedNeighs"):
        assert np.array_equal(a[k], b[k])
    assert np.array_equal(a["pdfs"], b["pdfs"])
