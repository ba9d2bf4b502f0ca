"""BASELINE.json configs at their stated sizes, HIP path (through the C-ABI) against the CPU oracle:

  cfg1  ModelNet40 MCClassS, 32 clouds x 1 024 points, grow 16   (models/MCClassS.py:29-71)
  cfg2  ModelNet40 MCClassH, 32 clouds x 4 096 points, 3 Poisson levels (models/MCClassH.py:30-187)
  cfg3  ShapeNet-Part MCSeg, 16 clouds x 8 192 points, grow 32, bf16 feature rows in the depth-wise layers (and again with f32 rows)
        (models/MCSeg.py:29-198; encoder, decoder and the two skip up-samplings)

For every level of the point hierarchy and every convolution of the graph: ALL integer outputs (keys, sort order, cell
tables, Poisson samples and their indices, CSR start indices, packed neighbours) bit-exact, KDE / convolution outputs
and the seven gradients within 1e-4 relative. The networks' dense layers (BN, 1x1 convs) are not on the hot path: each
convolution is fed seeded random features / out-gradients of the shape the graph gives it."""
import math

import numpy as np
import pytest

from tests.helpers import make_mlp, conv_nb, elementwise_excess
from mccnn_amd.workloads import modelnet_like, mcclass_s, mcclass_h, mcseg

pytestmark = pytest.mark.gpu
RTOL = 1e-4  # north_star: "fp32 features within 1e-4 rel"


def _wrap(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _unwrap(t):
    return t.detach().cpu().numpy()


def _ident(x):
    return x


@pytest.fixture(scope="module")
def oracle_omp():
    """The OpenMP build of the oracle: identical integer outputs, parameter gradients summed in double."""
    from oracle.oracle import Oracle
    return Oracle(omp=True)


def build_hierarchy(ops, wrap, unwrap, pts, bids, feats, B, radii):
    """PointHierarchy.__init__ (MCConvBuilder.py:101-128) op by op; returns per-level handles and the integer record."""
    P, Bi, F = wrap(pts), wrap(bids), wrap(feats)
    mn, mx = ops.compute_aabb(P, Bi, B, True)
    levels = [(P, Bi, F)]
    rec = [dict(aabbMin=unwrap(mn), aabbMax=unwrap(mx))]
    for r in radii:
        cP, cB, cF = levels[-1]
        keys, idx = ops.sort_points_step1(cP, cB, mn, mx, B, r, True)
        sP, sB, sF, cells = ops.sort_points_step2(cP, cB, cF, keys, idx, mn, mx, B, r, True)
        sp, sb, si = ops.poisson_sampling(sP, sB, cells, mn, mx, r, B, True)
        sf = ops.get_sampled_features(si, sF)
        ti = ops.transform_indexs(si, idx)
        levels.append((sp, sb, sf))
        rec.append(dict(keys=unwrap(keys), indexs=unwrap(idx), cellIndexs=unwrap(cells), samplePts=unwrap(sp),
                        sampleBatchs=unwrap(sb), sampleIndexs=unwrap(si), sampleFeatures=unwrap(sf),
                        transformedIndexs=unwrap(ti)))
    return mn, mx, levels, rec


def bf16_round(a):
    """float32 array -> the float32 values of its bfloat16 rounding (round to nearest even, what the kernels store)."""
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def run_conv(ops, wrap, unwrap, mn, mx, levels, B, spec, seed, is_gpu, bf16=False):
    """One create_convolution (MCConvBuilder.py:349-427): grid of the input level, neighbours of the output level's
    points, KDE, convolution forward and backward on seeded random features / out-gradients. bf16: the depth-wise
    layer keeps feature / output / gradient ROWS in bfloat16 (GPU) -- the oracle gets the same rounded values as
    float32 and its float32 results are rounded the same way by the caller."""
    lin, lout, radius, window, fin, fout, combin = spec[:7]
    inP, inB, _ = levels[lin]
    outP, outB, _ = levels[lout]
    n, m = int(inP.shape[0]), int(outP.shape[0])
    rng = np.random.default_rng(seed)
    feats = (2 * rng.random((n, fin)) - 1).astype(np.float32)
    outF = fout if combin else fin
    og = (2 * rng.random((m, outF)) - 1).astype(np.float32)
    if bf16:
        feats, og = bf16_round(feats), bf16_round(og)
    w = make_mlp(conv_nb(fin, fout, combin), seed + 1)
    keys, idx = ops.sort_points_step1(inP, inB, mn, mx, B, radius, True)
    sP, sB, sF, cells = ops.sort_points_step2(inP, inB, wrap(feats), keys, idx, mn, mx, B, radius, True)
    start, packed = ops.find_neighbors(outP, outB, sP, cells, mn, mx, radius, B, True)
    pdfs = ops.compute_pdf(sP, sB, mn, mx, start, packed, window, radius, B, True)
    r = dict(keys=unwrap(keys), indexs=unwrap(idx), cellIndexs=unwrap(cells), startIndexs=unwrap(start),
             packedNeighs=unwrap(packed), pdfs=unwrap(pdfs))
    if is_gpu:
        import torch
        tw = {k: wrap(v).requires_grad_(True) for k, v in w.items()}
        sFr = sF.detach().clone()
        if bf16:
            sFr = sFr.to(torch.bfloat16)      # exact: the values are bf16 already
        sFr.requires_grad_(True)
        out = ops.spatial_conv(sP, sFr, sB, pdfs, outP, start, packed, mn, mx, tw["w1"], tw["w2"], tw["w3"], tw["b1"],
                               tw["b2"], tw["b3"], fout, combin, B, radius, True, True)
        out.backward(wrap(og).to(out.dtype))
        r["out"] = unwrap(out.float())
        r["grads"] = [unwrap(t.float()) for t in (sFr.grad, tw["w1"].grad, tw["b1"].grad, tw["w2"].grad, tw["b2"].grad,
                                          tw["w3"].grad, tw["b3"].grad)]
    else:
        a = (sP, sF, sB, pdfs, outP, start, packed, mn, mx, w["w1"], w["w2"], w["w3"], w["b1"], w["b2"], w["b3"])
        r["out"] = ops.spatial_conv(*a, fout, combin, B, radius, True, True)
        r["grads"] = list(ops.spatial_conv_grad(*a, og, fout, combin, B, radius, True, True))
    return r


def rel_err(got, ref):
    """max |diff| / max |ref| (the caller holds it to RTOL); every ELEMENT is held to |d| <= RTOL |ref| + 1e-5 max |ref| here."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    ex = elementwise_excess(got, ref, RTOL)
    assert ex <= 1.0, "an element is %.2f x outside |d| <= 1e-4 |ref| + 1e-5 max|ref|" % ex
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)) if ref.size else 0.0


INT_HIER = ["keys", "indexs", "cellIndexs", "sampleBatchs", "sampleIndexs", "transformedIndexs"]
INT_CONV = ["keys", "indexs", "cellIndexs", "startIndexs", "packedNeighs"]
GRADS = ["featGrad", "dw1", "db1", "dw2", "db2", "dw3", "db3"]
S3 = math.sqrt(3.0) + 0.1

# (inLevel, outLevel, convRadius, KDEWindow, Fin, Fout, multiFeatureConv, bf16 rows): the graphs of mccnn_amd.workloads
# (each row cites its create_convolution call there), layers of identical shape over identical levels taken once
def _specs(convs):
    seen, out = set(), []
    for c in convs:
        t = (c.lin, c.lout, c.radius, c.window, c.fin, c.fout, c.combin, c.bf16)
        if t not in seen:
            seen.add(t)
            out.append(t)
    return out


MCCLASS_S_K16 = _specs(mcclass_s(16))  # models/MCClassS.py:38-71, grow 16 (BASELINE cfg1)
MCCLASS_H_K16 = _specs(mcclass_h(16))  # models/MCClassH.py:40-187, both logit branches, grow 16 (7 distinct of 10 layers)


def bf16_close(got, ref32):
    """got: values read back from bf16 storage; ref32: the oracle's float32 result. Both sides round nearly equal float32
    numbers (~1e-6 apart), so after rounding they are equal or ONE bf16 step (2^-8 of the element) apart."""
    ref = bf16_round(ref32)
    scale = np.abs(ref32).max()
    diff = np.abs(got.astype(np.float64) - ref)
    assert np.all(diff <= 2.0 ** -7 * np.abs(ref) + 1e-6 * scale), float((diff / np.maximum(np.abs(ref), 1e-6 * scale)).max())
    return float((diff > 0).mean())


def check_config(mc, orc, n_per, B, radii, convs, seed, level_sizes_strict=True):
    pts, bids = modelnet_like(n_per, B, seed)
    feats = np.ones((len(pts), 1), np.float32)  # ModelNet.py:168: one constant input feature
    gmn, gmx, glev, grec = build_hierarchy(mc, _wrap, _unwrap, pts, bids, feats, B, radii)
    omn, omx, olev, orec = build_hierarchy(orc, _ident, _ident, pts, bids, feats, B, radii)
    assert np.array_equal(grec[0]["aabbMin"], orec[0]["aabbMin"]) and np.array_equal(grec[0]["aabbMax"], orec[0]["aabbMax"])
    sizes = []
    for lvl in range(1, len(radii) + 1):
        for k in INT_HIER + ["samplePts", "sampleFeatures"]:
            assert grec[lvl][k].shape == orec[lvl][k].shape, (lvl, k, grec[lvl][k].shape, orec[lvl][k].shape)
            assert np.array_equal(grec[lvl][k], orec[lvl][k]), "level %d: %s differs" % (lvl, k)
        sizes.append(len(orec[lvl]["sampleIndexs"]))
    assert sizes[0] > sizes[1] > sizes[2] >= B  # a real three-level hierarchy; the last level holds >= 1 point per cloud
    worst = {}
    for ci, spec in enumerate(convs):
        bf16 = len(spec) > 7 and spec[7]
        g = run_conv(mc, _wrap, _unwrap, gmn, gmx, glev, B, spec, 100 + ci, True, bf16)
        o = run_conv(orc, _ident, _ident, omn, omx, olev, B, spec, 100 + ci, False, bf16)
        for k in INT_CONV:
            assert g[k].shape == o[k].shape, (spec, k, g[k].shape, o[k].shape)
            assert np.array_equal(g[k], o[k]), "conv %s: %s differs" % (spec, k)
        errs = {"pdfs": rel_err(g["pdfs"], o["pdfs"])}
        if bf16:  # rows stored in bf16: equal to the rounded oracle rows up to one bf16 step; parameter gradients are f32
            worst["bf16_rows_off_by_one_step"] = max(worst.get("bf16_rows_off_by_one_step", 0.0),
                                                     bf16_close(g["out"], o["out"]), bf16_close(g["grads"][0], o["grads"][0]))
        else:
            errs["out"] = rel_err(g["out"], o["out"])
        for nm, a, b in list(zip(GRADS, g["grads"], o["grads"]))[1 if bf16 else 0:]:
            errs[nm] = rel_err(a, b)
        for nm, e in errs.items():
            assert e <= RTOL, "conv %s: %s max |diff| / max |ref| = %.3e" % (spec, nm, e)
            worst[nm] = max(worst.get(nm, 0.0), e)
    return sizes, worst


def test_cfg0_uniform4096_conv_3to8_on_its_own_input(mc, oracle_omp):
    """BASELINE configs[0] as stated: ONE 4 096-point uniform cloud (seed 1), relative radius 0.1 (10^3 cells), one
    same-level convolution Fin = 3 -> Fout = 8 (nb = 3): grid, neighbours, KDE, forward and all seven gradients on that
    very input (mccnn_amd.workloads.CONFIGS['cfg0'] -- the input bench.py times for this config)."""
    from mccnn_amd.workloads import CONFIGS, config_points
    cfg = CONFIGS["cfg0"]
    pts, bids, B = config_points(cfg)
    assert pts.shape == (4096, 3) and B == 1 and len(cfg.convs) == 1
    c = cfg.convs[0]
    spec = (c.lin, c.lout, c.radius, c.window, c.fin, c.fout, c.combin)
    assert spec == (0, 0, 0.1, 0.2, 3, 8, True)
    feats = np.ones((len(pts), 1), np.float32)
    gmn, gmx, glev, grec = build_hierarchy(mc, _wrap, _unwrap, pts, bids, feats, B, [])
    omn, omx, olev, orec = build_hierarchy(oracle_omp, _ident, _ident, pts, bids, feats, B, [])
    assert np.array_equal(grec[0]["aabbMin"], orec[0]["aabbMin"]) and np.array_equal(grec[0]["aabbMax"], orec[0]["aabbMax"])
    g = run_conv(mc, _wrap, _unwrap, gmn, gmx, glev, B, spec, 100, True)
    o = run_conv(oracle_omp, _ident, _ident, omn, omx, olev, B, spec, 100, False)
    assert g["cellIndexs"].shape == (1, 10, 10, 10, 2)
    for k in INT_CONV:
        assert g[k].shape == o[k].shape and np.array_equal(g[k], o[k]), k
    errs = {"pdfs": rel_err(g["pdfs"], o["pdfs"]), "out": rel_err(g["out"], o["out"])}
    for nm, a, b in zip(GRADS, g["grads"], o["grads"]):
        errs[nm] = rel_err(a, b)
    print("cfg0: E = %d" % len(o["packedNeighs"]), {k: "%.1e" % v for k, v in errs.items()})
    for nm, e in errs.items():
        assert e <= RTOL, (nm, e)


def test_cfg1_mcclass_s_32x1024(mc, oracle_omp):
    sizes, worst = check_config(mc, oracle_omp, 1024, 32, [0.1, 0.4, S3], MCCLASS_S_K16, 41)
    print("cfg1 level sizes", sizes, "worst rel errs", {k: "%.1e" % v for k, v in worst.items()})


def test_cfg2_mcclass_h_32x4096(mc, oracle_omp):
    sizes, worst = check_config(mc, oracle_omp, 4096, 32, [0.1, 0.4, S3], MCCLASS_H_K16, 43)
    print("cfg2 level sizes", sizes, "worst rel errs", {k: "%.1e" % v for k, v in worst.items()})


MCSEG_K32 = _specs(mcseg(32))  # models/MCSeg.py:36-198, grow 32 (BASELINE cfg3: "bf16 features" -- bf16 feature storage in the depth-wise layers)
MCSEG_K32_F32 = _specs(mcseg(32, bf16=False))  # the same graph with the reference's own f32 rows


def test_cfg3_mcseg_16x8192_bf16_rows(mc, oracle_omp):
    sizes, worst = check_config(mc, oracle_omp, 8192, 16, [0.025, 0.1, 0.4], MCSEG_K32, 47)
    print("cfg3 level sizes", sizes, "worst", {k: "%.1e" % v for k, v in worst.items()})


def test_cfg3_mcseg_16x8192_f32_rows(mc, oracle_omp):
    sizes, worst = check_config(mc, oracle_omp, 8192, 16, [0.025, 0.1, 0.4], MCSEG_K32_F32, 47)
    print("cfg3 (f32 rows) level sizes", sizes, "worst", {k: "%.1e" % v for k, v in worst.items()})


def test_cfg4_mcsegscannet_room(mc, oracle_omp):
    """BASELINE cfg4's GRAPH (models/MCSegScanNet.py:29-249, mccnn_amd.workloads.mcseg_scannet(64)): one 100 000-point
    non-uniform room, hierarchy [0.1, 0.2, 0.4, 0.8] ABSOLUTE, all 17 convolutions -- Pool_* / Up_* cross levels, Up_1_3 and
    Up_1_4 skip levels, rows up to 512 features -- through the builder API on both sides: the product's default path
    (native step executor on the GPU) against the identical graph on the oracle ops (ops=, CPU tensors). Hierarchy and
    every grid / neighbour list bit-exact, PDFs, outputs and all seven gradients of every layer within 1e-4, norm-wise
    and per element."""
    import torch
    from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder
    from mccnn_amd.workloads import CONFIGS, config_points
    from tests.oracle_ops import OracleOps
    cfg = CONFIGS["cfg4"]
    pts, bids, B = config_points(cfg, 1)
    assert len(pts) == 100000 and B == 1 and not cfg.relative and list(cfg.hierarchy) == [0.1, 0.2, 0.4, 0.8]
    oo = OracleOps(oracle_omp)
    P, Bi = _wrap(pts), _wrap(bids)
    Pc, Bc = torch.from_numpy(pts), torch.from_numpy(bids)
    ph = PointHierarchy(P, torch.ones((len(pts), 1), device="cuda"), Bi, list(cfg.hierarchy), "PH", B, False)
    phc = PointHierarchy(Pc, torch.ones((len(pts), 1)), Bc, list(cfg.hierarchy), "PH", B, False, ops=oo)
    sizes = [int(p.shape[0]) for p in ph.points_]
    assert len(sizes) == 5 and sizes[0] == 100000 and all(a > b for a, b in zip(sizes, sizes[1:]))
    for l in range(1, 5):
        assert np.array_equal(_unwrap(ph.points_[l]), phc.points_[l].numpy()), l
        assert np.array_equal(_unwrap(ph.batchIds_[l]), phc.batchIds_[l].numpy()), l
        assert np.array_equal(_unwrap(ph.sampledIndexs_[l - 1]), phc.sampledIndexs_[l - 1].numpy()), l
    torch.manual_seed(17)
    cb = ConvolutionBuilder(KDEWindow=0.25, relativeRadius=False)
    cbc = ConvolutionBuilder(KDEWindow=0.25, relativeRadius=False, ops=oo)
    cb.reset()
    cbc.reset()
    assert cb.native_
    rng = np.random.default_rng(5)
    worst = {}
    for ci, c in enumerate(cfg.convs):
        n, m = sizes[c.lin], sizes[c.lout]
        outF = c.fout if c.combin else c.fin
        f = (2 * rng.random((n, c.fin)) - 1).astype(np.float32)
        og = (2 * rng.random((m, outF)) - 1).astype(np.float32)
        F = _wrap(f).requires_grad_(True)
        out = cb.create_convolution(c.name, ph, c.lin, F, c.fin, c.radius, ph, c.lout, c.combin, c.fout, c.window)
        names = [c.name + s for s in ("_weights", "_biases", "_weights2", "_biases2", "_weights3", "_biases3")]
        gp = dict(cb.named_parameters())
        cbc.load_state_dict({k: gp[k].detach().cpu().clone() for k in names}, strict=False)
        Fc = torch.from_numpy(f).requires_grad_(True)
        outc = cbc.create_convolution(c.name, phc, c.lin, Fc, c.fin, c.radius, phc, c.lout, c.combin, c.fout, c.window)
        cp = dict(cbc.named_parameters())
        g_gpu = torch.autograd.grad([out], [F] + [gp[k] for k in names], [_wrap(og)])
        g_cpu = torch.autograd.grad([outc], [Fc] + [cp[k] for k in names], [torch.from_numpy(og)])
        torch.cuda.synchronize()
        e = {"out": rel_err(_unwrap(out), outc.detach().numpy())}
        for nm, a, b in zip(GRADS, g_gpu, g_cpu):
            e[nm] = rel_err(_unwrap(a), b.numpy())
        for nm, v in e.items():
            assert v <= RTOL, "%s: %s max |diff| / max |ref| = %.3e" % (c.name, nm, v)
            worst[nm] = max(worst.get(nm, 0.0), v)
    assert len(cb.cacheGeo_) == len(cbc.cacheNeighs_) and list(cb.cacheNeighs_) == list(cbc.cacheNeighs_)
    assert list(cb.cacheGrids_) == list(cbc.cacheGrids_) and list(cb.cachePDFs_) == list(cbc.cachePDFs_)
    for k in cbc.cacheGrids_:     # sortPts, sortBatchs, cellIndexs, index_new_pos
        for a, b in zip(cb.cacheGrids_[k][:4], cbc.cacheGrids_[k][:4]):
            assert np.array_equal(_unwrap(a).reshape(-1), b.detach().numpy().reshape(-1)), k
    edges = 0
    for k in cbc.cacheNeighs_:
        (s0, p0), (s1, p1) = cb.cacheNeighs_[k], cbc.cacheNeighs_[k]
        assert np.array_equal(_unwrap(s0), s1.detach().numpy()) and np.array_equal(_unwrap(p0), p1.detach().numpy()), k
        edges += len(p1)
    for k in cbc.cachePDFs_:
        assert rel_err(_unwrap(cb.cachePDFs_[k].value()), cbc.cachePDFs_[k].detach().numpy()) <= RTOL, k
    print("cfg4 graph: levels", sizes, "lists", len(cbc.cacheNeighs_), "edges", edges, {k: "%.1e" % v for k, v in worst.items()})
