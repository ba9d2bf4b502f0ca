"""The native step executor (csrc/exec.hip, mccnn_amd/native.py: one library call per convolution geometry, one per layer
and direction) against the op-by-op chain of the same builder and against the oracle: same integer outputs bit for bit,
float outputs within the feature-path tolerance (they are the same kernels; only the row order of a plan may differ)."""
import numpy as np
import pytest

from tests.helpers import make_cloud, make_room, assert_float_close

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def _t(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _close(a, b, tol, what):
    """norm-wise and per element (tests/helpers.py)"""
    assert_float_close(a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy(), tol, what,
                       floor=1e-5 if tol <= 1e-3 else 1e-2)


LAYERS = [  # name, lin, lout, fin, fout, combin, radius, bf16
    ("Conv_f1", 0, 0, 1, 16, True, 0.12, False),
    ("Conv_3to8", 0, 0, 3, 8, True, 0.12, False),
    ("Conv_3to8_b", 0, 0, 3, 8, True, 0.12, False),    # second combin layer on the same list: deterministic feature gradient
    ("Conv_dw", 0, 0, 16, 16, False, 0.12, False),
    ("Pool_dw", 0, 1, 16, 16, False, 0.2, False),       # another output level: new list, own grid
    ("Pool_f1", 0, 1, 1, 8, True, 0.12, False),         # same grid as Conv_* (radius 0.12), another list: grid shared
    ("Conv_l1", 1, 1, 32, 32, False, 0.3, False),
    ("Up_dw", 1, 0, 16, 16, False, 0.3, False),         # same grid as Conv_l1, centres of level 0
    ("Conv_bf16", 0, 0, 16, 16, False, 0.12, True),
    ("Conv_2to5", 0, 0, 2, 5, True, 0.12, False),       # combin with padded neurons (10 -> 16)
]


def _run(mc, native, pts, bids, B, relative, feats, ogs, state=None, radius_scale=1.0):
    import torch
    from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder
    P, Bi = _t(pts), _t(bids)
    F0 = torch.ones((len(pts), 1), dtype=torch.float32, device="cuda")
    ph = PointHierarchy(P, F0, Bi, [0.1 * radius_scale], "PH", B, relative)
    torch.manual_seed(99)
    cb = ConvolutionBuilder(KDEWindow=0.25, relativeRadius=relative, native=native)
    if state is not None:
        cb.load_state_dict(state)
    cb.reset()
    outs, fts = [], []
    for (name, lin, lout, fin, fout, combin, radius, bf16) in LAYERS:
        n = ph.points_[lin].shape[0]
        f = feats.setdefault(name, (2 * torch.rand((n, fin), device="cuda") - 1))
        if bf16:
            f = f.to(torch.bfloat16)
        f = f.detach().clone().requires_grad_(True)
        fts.append(f)
        outs.append(cb.create_convolution(name, ph, lin, f, fin, radius * radius_scale, ph, lout, combin, fout))
    for o, (name, *_rest) in zip(outs, LAYERS):
        ogs.setdefault(name, (2 * torch.rand(o.shape, device="cuda") - 1))
    params = list(cb.parameters())
    grads = torch.autograd.grad(outs, fts + params, [ogs[l[0]].to(o.dtype) for l, o in zip(LAYERS, outs)])
    torch.cuda.synchronize()
    return cb, ph, outs, grads, [n_ for n_, _ in cb.named_parameters()]


@pytest.mark.parametrize("relative", [True, False], ids=["relative", "absolute"])
def test_native_executor_equals_the_op_chain(mc, relative):
    import torch
    pts, bids = make_cloud(3000, 3, 5, "clustered", True)
    feats, ogs = {}, {}
    scale = 1.0 if relative else 0.6
    cb0, ph0, outs0, grads0, names0 = _run(mc, False, pts, bids, 3, relative, feats, ogs, radius_scale=scale)
    assert not cb0.cacheGeo_
    sd = {k: v.detach().clone() for k, v in cb0.state_dict().items()}
    cb1, ph1, outs1, grads1, names1 = _run(mc, True, pts, bids, 3, relative, feats, ogs, state=sd, radius_scale=scale)
    assert len(cb1.cacheGeo_) == 5 and len(cb1.cacheGeoGrid_) == 3   # five lists over three grids
    assert names0 == names1
    assert list(cb0.cacheGrids_) == list(cb1.cacheGrids_) and list(cb0.cacheNeighs_) == list(cb1.cacheNeighs_)
    assert list(cb0.cachePDFs_) == list(cb1.cachePDFs_)
    # the geometry: same grids, lists and PDFs as the op chain's cache entries
    for k in cb0.cacheGrids_:
        for a, b in zip(cb0.cacheGrids_[k][:4], cb1.cacheGrids_[k][:4]):
            assert torch.equal(a.reshape(-1), b.reshape(-1)), k
    for k in cb0.cacheNeighs_:
        (s0, p0), (s1, p1) = cb0.cacheNeighs_[k], cb1.cacheNeighs_[k]
        assert torch.equal(s0, s1) and torch.equal(p0, p1), k
    for k in cb0.cachePDFs_:
        assert torch.equal(cb0.cachePDFs_[k], cb1.cachePDFs_[k].value()), k
    for l, a, b in zip(LAYERS, outs0, outs1):
        _close(b, a, 2e-5 if not l[7] else 1e-2, l[0])
    for i, (a, b) in enumerate(zip(grads0, grads1)):
        what = LAYERS[i][0] + ":featGrad" if i < len(LAYERS) else names0[i - len(LAYERS)]
        bf = i < len(LAYERS) and LAYERS[i][7]
        _close(b, a, 2e-5 if not bf else 1e-2, what)


def test_native_capacity_overflow_and_changing_batches(mc, oracle):
    """First batch of a shape: the neighbour list is sized by a plain guess; a list that does not fit is rebuilt with
    the exact size (dense cloud: ~200 neighbours per point against the guess of 48). Later batches of other sizes reuse
    the edges-per-centre ratio. Every batch is checked against the oracle's list."""
    import torch
    from mccnn_amd import native
    from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder
    native._EDGE_GUESS.clear()
    native._EDGE_RATIO.clear()
    cb = ConvolutionBuilder(KDEWindow=0.25, relativeRadius=True, native=True)
    torch.manual_seed(3)
    for n_per, seed in ((900, 1), (1100, 2), (500, 3)):
        pts, bids = make_cloud(n_per, 2, seed, "uniform", False)
        P, Bi = _t(pts), _t(bids)
        F = (2 * torch.rand((len(pts), 2), device="cuda") - 1).requires_grad_(True)
        ph = PointHierarchy(P, F, Bi, [], "PH", 2, True)
        cb.reset()
        out = cb.create_convolution("Conv", ph, 0, F, 2, 0.45, outNumFeatures=4, multiFeatureConv=True)
        out.sum().backward()
        geo = next(iter(cb.cacheGeo_.values()))
        mn, mx = oracle.compute_aabb(pts, bids, 2, True)
        k_, i_ = oracle.sort_points_step1(pts, bids, mn, mx, 2, 0.45, True)
        sp, sb, _, cl = oracle.sort_points_step2(pts, bids, np.zeros((len(pts), 1), np.float32), k_, i_, mn, mx, 2, 0.45, True)
        st, pk = oracle.find_neighbors(pts, bids, sp, cl, mn, mx, 0.45, 2, True)
        gs, gp = cb.cacheNeighs_["PH|0|0.45|True|PH|0"]
        assert geo.e == len(pk) and geo.e <= geo.e_cap
        assert np.array_equal(gs.cpu().numpy(), st) and np.array_equal(gp.cpu().numpy(), pk)
        if seed == 1:
            assert len(pk) > 48 * len(pts) + 1024   # the first guess could not hold it: the rebuild ran
    torch.cuda.synchronize()


def test_native_room_layers_against_the_oracle(mc, oracle_omp):
    """The three layer shapes of the headline workload on a 20k-point room through the native executor, against the
    OpenMP oracle: outputs and all seven gradients."""
    import torch
    from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder
    from tests.helpers import make_mlp, conv_nb
    pts = make_room(20000, 20180601)
    bids = np.zeros((len(pts), 1), np.int32)
    B, R = 1, 0.1
    P, Bi = _t(pts), _t(bids)
    orc = oracle_omp
    mn, mx = orc.compute_aabb(pts, bids, B, False)
    k_, i_ = orc.sort_points_step1(pts, bids, mn, mx, B, R, False)
    rng = np.random.default_rng(2)
    for name, fin, fout, combin in (("1to64", 1, 64, True), ("3to8", 3, 8, True), ("dw64", 64, 64, False)):
        feats = (2 * rng.random((len(pts), fin)) - 1).astype(np.float32)
        sp, sb, sf, cl = orc.sort_points_step2(pts, bids, feats, k_, i_, mn, mx, B, R, False)
        st, pk = orc.find_neighbors(pts, bids, sp, cl, mn, mx, R, B, False)
        pdf = orc.compute_pdf(sp, sb, mn, mx, st, pk, 0.2, R, B, False)
        w = make_mlp(conv_nb(fin, fout, combin), 7)
        outF = fout if combin else fin
        og = (2 * rng.random((len(pts), outF)) - 1).astype(np.float32)
        args = (sp, sf, sb, pdf, pts, st, pk, mn, mx, w["w1"], w["w2"], w["w3"], w["b1"], w["b2"], w["b3"])
        ref = orc.spatial_conv(*args, fout, combin, B, R, False, True)
        rg = orc.spatial_conv_grad(*args, og, fout, combin, B, R, False, True)
        fg_ref = orc.sort_points_step2_grad(i_, np.zeros_like(sp), rg[0])[1]
        F = _t(feats).requires_grad_(True)
        ph = PointHierarchy(P, F, Bi, [], "PH", B, False)
        cb = ConvolutionBuilder(KDEWindow=0.2, relativeRadius=False, native=True)
        nb = conv_nb(fin, fout, combin)
        sd = {"C_weights": _t(w["w1"]).reshape(3, 8 * nb), "C_biases": _t(w["b1"]), "C_weights2": _t(w["w2"]).reshape(nb, 8, 8),
              "C_biases2": _t(w["b2"]).reshape(nb, 8), "C_weights3": _t(w["w3"]).reshape(nb, 8, 8),
              "C_biases3": _t(w["b3"]).reshape(nb, 8)}
        cb.load_state_dict(sd)
        cb.reset()
        out = cb.create_convolution("C", ph, 0, F, fin, R, outNumFeatures=fout, multiFeatureConv=combin)
        assert cb.cacheGeo_
        out.backward(_t(og))
        torch.cuda.synchronize()
        gs, gp = cb.cacheNeighs_["PH|0|%s|False|PH|0" % R]
        assert np.array_equal(gs.cpu().numpy(), st) and np.array_equal(gp.cpu().numpy(), pk)
        _close(out, torch.from_numpy(ref), RTOL, name + ":out")
        _close(F.grad, torch.from_numpy(fg_ref), RTOL, name + ":featGrad")
        named = dict(cb.named_parameters())
        for key, r in (("C_weights", rg[1]), ("C_biases", rg[2]), ("C_weights2", rg[3]), ("C_biases2", rg[4]),
                       ("C_weights3", rg[5]), ("C_biases3", rg[6])):
            _close(named[key].grad.reshape(-1), torch.from_numpy(np.asarray(r).reshape(-1)), RTOL, name + ":" + key)


def test_native_path_keeps_the_ops_shape_rules(mc):
    """spatial_conv.cc:290-296: the output neurons must be a multiple of the input features, and a depth-wise layer needs
    exactly numInFeatures neurons (numInFeatures % 8 == 0) -- the native path raises like the op does."""
    import torch
    from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder
    from mccnn_amd.MCConvModule import InvalidArgumentError
    pts, bids = make_cloud(500, 1, 3, "uniform", False)
    P, Bi = _t(pts), _t(bids)
    ph = PointHierarchy(P, torch.ones((len(pts), 1), device="cuda"), Bi, [], "PH", 1, True)
    for fin, fout, combin in ((4, 4, False), (3, 5, True)):
        for native in (True, False):
            cb = ConvolutionBuilder(relativeRadius=True, native=native)
            F = torch.rand((len(pts), fin), device="cuda")
            with pytest.raises(InvalidArgumentError):
                cb.create_convolution("C", ph, 0, F, fin, 0.2, outNumFeatures=fout, multiFeatureConv=combin)


def test_learned_geometry_prefetch_on_side_streams(mc):
    """From the second step on the builder issues ALL geometries of the step at its first create_convolution, on side
    streams (the list is learned from the previous step). Same outputs and gradients as the first (inline) step bit for
    bit -- also when the batch changes between steps and when a step asks for a geometry the plan does not hold."""
    import torch
    from mccnn_amd import native
    from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder
    if not native.side_streams_available():
        pytest.skip("torch extension not built")
    layers = [l for l in LAYERS if not l[7]]
    clouds = [make_cloud(2500, 3, s, "clustered", True) for s in (5, 6)]
    torch.manual_seed(1)
    cb = ConvolutionBuilder(KDEWindow=0.25, relativeRadius=True)
    feats, ogs = {}, {}

    def step(ci, extra=False):
        pts, bids = clouds[ci]
        P, Bi = _t(pts), _t(bids)
        ph = PointHierarchy(P, torch.ones((len(pts), 1), device="cuda"), Bi, [0.1], "PH", 3, True)
        cb.reset()
        outs, fts = [], []
        for (name, lin, lout, fin, fout, combin, radius, _bf) in layers:
            n = ph.points_[lin].shape[0]
            f = feats.setdefault((ci, name), 2 * torch.rand((n, fin), device="cuda") - 1).detach().clone().requires_grad_(True)
            fts.append(f)
            outs.append(cb.create_convolution(name, ph, lin, f, fin, radius, ph, lout, combin, fout))
        if extra:   # a geometry no earlier step asked for
            f = feats.setdefault((ci, "extra"), 2 * torch.rand((ph.points_[1].shape[0], 8), device="cuda") - 1).detach().clone().requires_grad_(True)
            fts.append(f)
            outs.append(cb.create_convolution("Extra", ph, 1, f, 8, 0.45, ph, 1, False, 8))
        for k, o in enumerate(outs):
            ogs.setdefault((ci, k), 2 * torch.rand(o.shape, device="cuda") - 1)
        grads = torch.autograd.grad(outs, fts + list(cb.parameters()), [ogs[(ci, k)] for k in range(len(outs))], allow_unused=True)
        sides = [g.core.side for g in cb.cacheGeo_.values()]
        torch.cuda.synchronize()
        return [o.detach().clone() for o in outs], [g.detach().clone() for g in grads if g is not None], sides

    ref0 = step(0)
    assert all(s < 0 for s in ref0[2])                  # nothing to learn from yet: built where they were asked for
    ref1 = step(1)
    # learned: issued on side streams, several of them (a list that outgrew the capacity guessed from the other batch is
    # built again where it is used: the first time a shape is seen)
    assert sum(1 for s in ref1[2] if s >= 0) >= 3 and len(set(s for s in ref1[2] if s >= 0)) > 1
    for rep in range(3):
        for ci, ref in ((0, ref0), (1, ref1)):
            got = step(ci)
            assert all(s >= 0 for s in got[2])
            for a, b in zip(got[0], ref[0]):
                assert torch.equal(a, b)
            for a, b in zip(got[1], ref[1]):   # (feature gradients of one-feature layers are summed with float atomics)
                assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
    # a step with one more geometry than the plan holds, then the plain graph again
    got = step(0, extra=True)
    assert sum(1 for s in got[2] if s < 0) == 1
    for a, b in zip(got[0][:len(ref0[0])], ref0[0]):
        assert torch.equal(a, b)
    got = step(1)
    for a, b in zip(got[0], ref1[0]):
        assert torch.equal(a, b)


def test_backward_detects_inputs_modified_in_place(mc):
    """The extension's autograd node holds its inputs as plain tensors and checks their versions itself: an in-place update
    of a kernel-MLP variable between forward and backward is an error, as with any autograd function; retain_graph works."""
    import torch
    import mccnn_amd.MCConvBuilder as MB
    from mccnn_amd import native
    from tests.helpers import make_cloud
    if not native.side_streams_available():
        pytest.skip("torch extension not built")
    pts, bids = make_cloud(2000, 2, 5, "uniform", True)
    P, Bi = torch.from_numpy(pts).cuda(), torch.from_numpy(bids).cuda()
    ph = MB.PointHierarchy(P, torch.ones((len(pts), 1), device="cuda"), Bi, [], "PH", 2)
    b = MB.ConvolutionBuilder(KDEWindow=0.2)
    f = torch.rand((len(pts), 1), device="cuda", requires_grad=True)
    out = b.create_convolution("c", ph, 0, f, 1, 0.15, outNumFeatures=8, multiFeatureConv=True)
    g1 = torch.autograd.grad(out.sum(), [f], retain_graph=True)[0]
    g2 = torch.autograd.grad(out.sum(), [f], retain_graph=True)[0]
    assert torch.allclose(g1, g2, rtol=1e-5, atol=1e-7)
    with torch.no_grad():
        next(iter(b.parameters())).mul_(2.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        out.sum().backward()


def test_geometries_dropped_before_their_totals_arrive(mc, oracle):
    """A geometry destroyed right after being queued (a prefetch nobody uses, an exception path): its pinned slot must not go
    back to the pool while the count pass can still write it -- a later geometry armed with -1 would read a stale total and
    convolve over a truncated list (torch_ext.cpp: parked slots). Hundreds of builds on side streams and on the caller's
    stream are dropped unread, then a fresh geometry of ANOTHER size must still report its own edge total."""
    import torch
    from mccnn_amd import native
    from mccnn_amd import MCConvModule as M
    sides = (-1, 0) if native._EXT is None else None   # (ctypes binding: every build runs on the caller's stream)
    clouds = []
    for n_per, seed in ((3000, 1), (700, 2), (1500, 3)):
        pts, bids = make_cloud(n_per, 2, seed, "clustered", False)
        P, Bi = _t(pts), _t(bids)
        mn, mx = M.compute_aabb(P, Bi, 2, True)
        clouds.append((pts, bids, P, Bi, mn, mx))
    radius = 0.2
    nc = M._num_cells(clouds[0][4], clouds[0][5], 2, radius, True)
    want = []
    for pts, bids, P, Bi, mn, mx in clouds:
        omn, omx = oracle.compute_aabb(pts, bids, 2, True)
        k_, i_ = oracle.sort_points_step1(pts, bids, omn, omx, 2, radius, True)
        sp, sb, _, cl = oracle.sort_points_step2(pts, bids, np.zeros((len(pts), 1), np.float32), k_, i_, omn, omx, 2, radius, True)
        want.append(len(oracle.find_neighbors(pts, bids, sp, cl, omn, omx, radius, 2, True)[1]))
    for rnd in range(60):
        for ci, (pts, bids, P, Bi, mn, mx) in enumerate(clouds):
            for side in (sides or (-1, rnd % 4)):
                g = native.build_geometry(P, Bi, P, Bi, mn, mx, 2, nc, radius, True, 0.25, True, side=side, fork=side >= 0)
                del g   # dropped before anybody asked for its total
        ci = rnd % 3
        pts, bids, P, Bi, mn, mx = clouds[ci]
        g = native.build_geometry(P, Bi, P, Bi, mn, mx, 2, nc, radius, True, 0.25, True)
        assert g.edges() == want[ci], (rnd, ci)
    torch.cuda.synchronize()


def test_geometries_dropped_before_their_totals_arrive_ctypes_binding():
    """The same through the ctypes binding of the C-ABI (MCCNN_TORCH_EXT=0: mccnn_amd.native parks the pinned words itself)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MCCNN_TORCH_EXT="0")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_native.py"), "-m", "gpu", "-x", "-q",
                          "-k", "test_geometries_dropped_before_their_totals_arrive and not ctypes"], env=env, capture_output=True,
                         text=True, timeout=600, cwd=root)
    assert out.returncode == 0 and "1 passed" in out.stdout, out.stdout[-1500:] + out.stderr[-500:]


@pytest.mark.parametrize("native", [True, False], ids=["native", "opchain"])
def test_use_pdf_false_against_the_oracle(mc, oracle, native):
    """usePDF=False (utils/MCConvBuilder.py:379-391: the PDF tensor is a tensor of ones, same cache key scheme): the
    builder's native executor (and its op-by-op path) against the IDENTICAL graph on the oracle ops -- a combin Fin = 1
    layer, a small-Fin combin layer, a depth-wise layer, a pooling layer to another level, and a layer that asks for
    usePDF=True on the same list (another cache entry, real densities). Lists bit-exact, outputs and all seven gradients
    within 1e-4 norm-wise and per element."""
    import torch
    from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder
    from tests.oracle_ops import OracleOps
    pts, bids = make_cloud(1500, 2, 11, "clustered", True)
    B = 2
    oo = OracleOps(oracle)
    P, Bi = _t(pts), _t(bids)
    ph = PointHierarchy(P, torch.ones((len(pts), 1), device="cuda"), Bi, [0.15], "PH", B, True)
    phc = PointHierarchy(torch.from_numpy(pts), torch.ones((len(pts), 1)), torch.from_numpy(bids), [0.15], "PH", B, True, ops=oo)
    assert np.array_equal(ph.points_[1].cpu().numpy(), phc.points_[1].numpy())
    torch.manual_seed(5)
    cb = ConvolutionBuilder(KDEWindow=0.25, relativeRadius=True, usePDF=False, native=native)
    cbc = ConvolutionBuilder(KDEWindow=0.25, relativeRadius=True, usePDF=False, ops=oo)
    cb.reset()
    cbc.reset()
    assert bool(cb.native_) == native
    rng = np.random.default_rng(3)
    layers = [  # name, lin, lout, fin, fout, combin, radius, usePDF override
        ("NP_f1", 0, 0, 1, 16, True, 0.2, None),
        ("NP_3to8", 0, 0, 3, 8, True, 0.2, None),
        ("NP_dw", 0, 0, 16, 16, False, 0.2, None),
        ("NP_pool", 0, 1, 8, 8, False, 0.3, None),
        ("WP_dw", 0, 0, 16, 16, False, 0.2, True),
    ]
    sizes = [int(p.shape[0]) for p in ph.points_]
    for (name, lin, lout, fin, fout, combin, radius, up) in layers:
        n, m = sizes[lin], sizes[lout]
        outF = fout if combin else fin
        f = (2 * rng.random((n, fin)) - 1).astype(np.float32)
        og = (2 * rng.random((m, outF)) - 1).astype(np.float32)
        F = _t(f).requires_grad_(True)
        out = cb.create_convolution(name, ph, lin, F, fin, radius, ph, lout, combin, fout, usePDF=up)
        names = [name + s for s in ("_weights", "_biases", "_weights2", "_biases2", "_weights3", "_biases3")]
        gp = dict(cb.named_parameters())
        cbc.load_state_dict({k: gp[k].detach().cpu().clone() for k in names}, strict=False)
        Fc = torch.from_numpy(f).requires_grad_(True)
        outc = cbc.create_convolution(name, phc, lin, Fc, fin, radius, phc, lout, combin, fout, usePDF=up)
        cp = dict(cbc.named_parameters())
        g_gpu = torch.autograd.grad([out], [F] + [gp[k] for k in names], [_t(og)])
        g_cpu = torch.autograd.grad([outc], [Fc] + [cp[k] for k in names], [torch.from_numpy(og)])
        torch.cuda.synchronize()
        _close(out, outc, RTOL, name + ":out")
        for nm, a, b in zip(["featGrad"] + names, g_gpu, g_cpu):
            _close(a, b, RTOL, name + ":" + nm)
    # two PDF cache entries over the radius-0.2 list: "...|False" (ones) and "...|True" (densities)
    assert list(cb.cachePDFs_) == list(cbc.cachePDFs_) and list(cb.cacheNeighs_) == list(cbc.cacheNeighs_)
    keys = [k for k in cbc.cachePDFs_ if k.endswith("|False")]
    assert len(keys) == 2 and len(cbc.cachePDFs_) == 3
    for k in cbc.cachePDFs_:
        a = cb.cachePDFs_[k]
        a = a.value() if hasattr(a, "value") else a
        if k.endswith("|False"):
            assert float(a.min()) == 1.0 and float(a.max()) == 1.0, k
        _close(a.reshape(-1), cbc.cachePDFs_[k].reshape(-1), RTOL, k)
    for k in cbc.cacheNeighs_:
        (s0, p0), (s1, p1) = cb.cacheNeighs_[k], cbc.cacheNeighs_[k]
        assert np.array_equal(s0.cpu().numpy(), s1.numpy()) and np.array_equal(p0.cpu().numpy(), p1.numpy()), k


@pytest.mark.parametrize("n_per,B,radius,relative,empty", [
    (2048, 1, 0.11, True, False),      # exactly the single-workgroup limit (grid_small_all)
    (2049, 1, 0.11, True, False),      # one past it: keys_hist / scan / park_ids / rank_move
    (700, 5, 0.3, True, True),         # clouds WITHOUT points in the batch (their cells stay (0, 0)); few cells, LDS histogram
    (4096, 1, 0.05, True, False),      # 8 000 cells, 4 096 centres: the in-fill prefix sum at its limit
    (4097, 1, 0.05, True, False),      # ... and one past it (stand-alone scan)
    (1500, 3, 0.07, False, False),     # absolute radius, whole-batch box
    (33000, 1, 0.02, True, False),     # aabb in three launches, 125 000 cells (multi-tile chained scan over the cells)
])
def test_fused_grid_and_search_chain_equals_the_ops(mc, oracle, n_per, B, radius, relative, empty):
    """Round 6's launch chain -- grid build as keys_hist / scan / park_ids / rank_move (cell table read off the prefix sum) or
    one workgroup, the visiting order, count -> (scan ->) fill, aabb in one launch -- through the native executor's geometry
    against the op-by-op surface AND the oracle: sorted points, batch ids, cell table, permutation, startIdx, packed: bit-exact."""
    import torch
    from mccnn_amd import native
    pts, bids = make_cloud(n_per, B, 21, "clustered", True)
    if empty:
        keep = (bids[:, 0] != 1) & (bids[:, 0] != 3)      # clouds 1 and 3 have no points
        pts, bids = np.ascontiguousarray(pts[keep]), np.ascontiguousarray(bids[keep])
    if not relative:
        pts = pts * 2.0
    P, Bi = _t(pts), _t(bids)
    mn, mx = mc.compute_aabb(P, Bi, B, relative)
    omn, omx = oracle.compute_aabb(pts, bids, B, relative)
    assert np.array_equal(mn.cpu().numpy(), omn) and np.array_equal(mx.cpu().numpy(), omx)
    # the op surface
    keys, idx = mc.sort_points_step1(P, Bi, mn, mx, B, radius, relative)
    sP, sB, _sF, cells = mc.sort_points_step2(P, Bi, torch.ones((len(pts), 1), device="cuda"), keys, idx, mn, mx, B, radius, relative)
    st, pk = mc.find_neighbors(P, Bi, sP, cells, mn, mx, radius, B, relative)
    # the native geometry (one library call)
    nc = mc._num_cells(mn, mx, B, radius, relative)
    geo = native.build_geometry(P, Bi, P, Bi, mn, mx, B, nc, radius, relative, 0.25, True)
    gP, gB, gC, gI, gInv = geo.grid()
    gst, gpk = geo.neighbors()
    torch.cuda.synchronize()
    assert torch.equal(gP, sP) and torch.equal(gB.reshape(-1), sB.reshape(-1)) and torch.equal(gC, cells) and torch.equal(gI, idx)
    assert torch.equal(gInv[idx.long()].cpu(), torch.arange(len(pts), dtype=torch.int32))
    assert torch.equal(gst.reshape(-1), st.reshape(-1)) and torch.equal(gpk, pk)
    # ... and the oracle
    ok_, oi_ = oracle.sort_points_step1(pts, bids, omn, omx, B, radius, relative)
    osP, osB, _f, ocl = oracle.sort_points_step2(pts, bids, np.zeros((len(pts), 1), np.float32), ok_, oi_, omn, omx, B, radius, relative)
    ost, opk = oracle.find_neighbors(pts, bids, osP, ocl, omn, omx, radius, B, relative)
    assert np.array_equal(gI.cpu().numpy(), oi_) and np.array_equal(gC.cpu().numpy().reshape(-1), ocl.reshape(-1))
    assert np.array_equal(gst.cpu().numpy().reshape(-1), ost.reshape(-1)) and np.array_equal(gpk.cpu().numpy(), opk)
    if empty:
        c = gC.cpu().numpy()
        assert not c[1].any() and not c[3].any()


def test_foreign_centres_get_a_visiting_order_without_changing_the_list(mc):
    """Centres that are not the gridded points (>= 16 384 of them: pooling / up-sampling lists) are searched in a
    cell-coherent order of their own -- keys_hist / scan / park_ids, arrival order inside a cell (round 6: no stable sort).
    The order is speed only: startIdx / packed equal the op surface's, whatever the arrival order was (two builds)."""
    import torch
    from mccnn_amd import native
    pts, bids = make_cloud(9000, 4, 33, "clustered", True)
    B, radius = 4, 0.08
    rng = np.random.default_rng(8)
    pick = np.sort(rng.choice(len(pts), 20000, replace=False))   # (clouds stay contiguous)
    cen = np.ascontiguousarray(pts[pick] + rng.normal(0, 1e-3, (len(pick), 3)).astype(np.float32))
    cb = np.ascontiguousarray(bids[pick])
    perm = np.concatenate([rng.permutation(np.nonzero(cb[:, 0] == b)[0]) for b in range(B)])   # scattered inside each cloud
    cen, cb = np.ascontiguousarray(cen[perm]), np.ascontiguousarray(cb[perm])
    P, Bi, Cn, Cb = _t(pts), _t(bids), _t(cen), _t(cb)
    mn, mx = mc.compute_aabb(P, Bi, B, True)
    keys, idx = mc.sort_points_step1(P, Bi, mn, mx, B, radius, True)
    sP, sB, _sF, cells = mc.sort_points_step2(P, Bi, torch.ones((len(pts), 1), device="cuda"), keys, idx, mn, mx, B, radius, True)
    st, pk = mc.find_neighbors(Cn, Cb, sP, cells, mn, mx, radius, B, True)
    nc = mc._num_cells(mn, mx, B, radius, True)
    for _ in range(2):
        geo = native.build_geometry(P, Bi, Cn, Cb, mn, mx, B, nc, radius, True, 0.25, True)
        gst, gpk = geo.neighbors()
        torch.cuda.synchronize()
        assert torch.equal(gst.reshape(-1), st.reshape(-1)) and torch.equal(gpk, pk)
    assert pk.shape[0] > 0


def test_batched_geometries_equal_the_single_chains(mc):
    """mccnn_geometry_build_batch (one launch per kernel kind over all geometries of a step; what prefetch_step issues) against
    mccnn_geometry_build of every geometry on its own: grids, cell tables, lists, PDFs bit-identical -- own grids, a shared
    grid (owner earlier in the batch), foreign centres with a visiting order of their own, a tiny and a large list, a list
    without densities. Also: 18 requests (two chunks)."""
    import ctypes as C
    import torch
    from mccnn_amd import _lib
    from mccnn_amd._lib import ptr, check
    lib = _lib.load()
    pts, bids = make_cloud(6000, 4, 41, "clustered", True)
    B = 4
    P, Bi = _t(pts), _t(bids)
    rng = np.random.default_rng(2)
    pick = np.sort(rng.choice(len(pts), 17000, replace=False))
    Cn, Cb = _t(np.ascontiguousarray(pts[pick])), _t(np.ascontiguousarray(bids[pick]))
    small = np.sort(rng.choice(len(pts), 300, replace=False))
    Sn, Sb = _t(np.ascontiguousarray(pts[small])), _t(np.ascontiguousarray(bids[small]))
    mn, mx = mc.compute_aabb(P, Bi, B, True)

    class Req(C.Structure):
        _fields_ = [("geometry", C.c_void_p), ("pts", C.c_void_p), ("batch_ids", C.c_void_p), ("n", C.c_int),
                    ("centres", C.c_void_p), ("centre_batch_ids", C.c_void_p), ("m", C.c_int), ("aabb_min", C.c_void_p),
                    ("aabb_max", C.c_void_p), ("batch_size", C.c_int), ("num_cells", C.c_int), ("radius", C.c_float),
                    ("scale_inv", C.c_int), ("window", C.c_float), ("use_pdf", C.c_int), ("e_capacity", C.c_int),
                    ("grid_from", C.c_void_p), ("buffer", C.c_void_p), ("buffer_bytes", C.c_size_t), ("total_host", C.c_void_p)]

    lib.mccnn_geometry_create.restype = C.c_void_p
    specs = [  # (points, ids, centres, ids, radius, use_pdf, grid owner index or None)
        (P, Bi, P, Bi, 0.08, 1, None),      # same level, own grid
        (P, Bi, Cn, Cb, 0.08, 1, 0),        # foreign centres (>= 16 384: visiting order), grid shared with request 0
        (P, Bi, Sn, Sb, 0.2, 1, None),      # few centres, own grid
        (Sn, Sb, Sn, Sb, 0.5, 0, None),     # tiny level, no densities
        (Cn, Cb, P, Bi, 0.05, 1, None),     # own grid over the centre set, the points as foreign centres
    ]
    specs = specs + [specs[2]] * 13         # 18 requests: more than one chunk
    def build(batched):
        outs, keep = [], []
        handles = [lib.mccnn_geometry_create() for _ in specs]
        reqs = (Req * len(specs))()
        slots = torch.empty(len(specs), dtype=torch.int32).pin_memory()
        for k, (p, b, c, cb, r, up, owner) in enumerate(specs):
            n, m = p.shape[0], c.shape[0]
            nc = mc._num_cells(mn, mx, B, r, True)
            cap = 900 * m
            nbytes = lib.mccnn_geometry_bytes(n, m, B, nc, cap, 0 if owner is not None else 1)
            buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            keep.append(buf)
            reqs[k] = Req(handles[k], ptr(p), ptr(b), n, ptr(c), ptr(cb), m, ptr(mn), ptr(mx), B, nc, r, 1, 0.25, up, cap,
                          handles[owner] if owner is not None else None, buf.data_ptr(), nbytes, slots.data_ptr() + 4 * k)
        st = torch.cuda.current_stream().cuda_stream
        if batched:
            check(lib.mccnn_geometry_build_batch(C.byref(reqs), len(specs), C.c_void_p(st)), "geometry_build_batch")
        else:
            for k, q in enumerate(reqs):
                check(lib.mccnn_geometry_build(C.c_void_p(q.geometry), C.c_void_p(q.pts), C.c_void_p(q.batch_ids), q.n, C.c_void_p(q.centres),
                                               C.c_void_p(q.centre_batch_ids), q.m, C.c_void_p(q.aabb_min), C.c_void_p(q.aabb_max), q.batch_size,
                                               q.num_cells, C.c_float(q.radius), q.scale_inv, C.c_float(q.window), q.use_pdf, q.e_capacity,
                                               C.c_void_p(q.grid_from) if q.grid_from else None, C.c_void_p(q.buffer), q.buffer_bytes,
                                               C.c_void_p(q.total_host), C.c_void_p(st)), "geometry_build")
        torch.cuda.synchronize()
        info = (C.c_longlong * 16)()
        for k, (p, b, c, cb, r, up, owner) in enumerate(specs):
            e = int(slots[k])
            assert 0 < e <= reqs[k].e_capacity, (k, e)
            check(lib.mccnn_geometry_info(C.c_void_p(handles[k]), info), "geometry_info")
            n, m, nc = p.shape[0], c.shape[0], reqs[k].num_cells
            def view(addr, count, dt):
                for t in keep:   # (the grid arrays of a geometry that shares a grid lie in the owner's buffer)
                    off = addr - t.data_ptr()
                    if 0 <= off < t.numel():
                        nb = count * torch.empty(0, dtype=dt).element_size()
                        return t[off:off + nb].view(dt).cpu()
                raise AssertionError("address outside the buffers")
            outs.append((e, view(info[0], n * 3, torch.float32), view(info[2], B * nc ** 3 * 2, torch.int32), view(info[3], n, torch.int32),
                         view(info[5], m, torch.int32), view(info[6], e * 2, torch.int32), view(info[7], e, torch.float32)))
        for h in handles:
            lib.mccnn_geometry_destroy(C.c_void_p(h))
        return outs
    lib.mccnn_geometry_build.argtypes = None
    a, b_ = build(False), build(True)
    for k, (x, y) in enumerate(zip(a, b_)):
        assert x[0] == y[0], k
        for t0, t1 in zip(x[1:], y[1:]):
            assert torch.equal(t0, t1), k
    assert a[3][6].min() == 1.0 and a[3][6].max() == 1.0   # usePDF = 0: ones


@pytest.mark.gpu
def test_batched_pieces_equal_the_single_chains(mc):
    """mccnn_geometry_prebuild_batch (the row plans and transposed lists of a step's small lists: one launch per kernel kind
    -- transposition chain or single-workgroup transposition, layout, slot fill) against mccnn_geometry_prebuild of every
    geometry on its own: the attached buffers are byte-identical. Lists: tiny (single-workgroup transposition), small plan
    over a list whose transposition is a chain, a large one (own chain behind the batch), a list without densities; masks
    1|2|4, 2 alone, 4 alone, 1 alone; both `avg` flags; 14 geometries (two flushes)."""
    import ctypes as C
    import torch
    from mccnn_amd import _lib
    from mccnn_amd._lib import ptr, check
    lib = _lib.load()
    lib.mccnn_geometry_create.restype = C.c_void_p
    lib.mccnn_geometry_build.argtypes = None
    lib.mccnn_geometry_prebuild_batch_ws_bytes.restype = C.c_size_t
    pts, bids = make_cloud(5000, 3, 43, "clustered", True)
    B = 3
    P, Bi = _t(pts), _t(bids)
    rng = np.random.default_rng(5)
    def sub(k):
        pick = np.sort(rng.choice(len(pts), k, replace=False))
        return _t(np.ascontiguousarray(pts[pick])), _t(np.ascontiguousarray(bids[pick]))
    Mn, Mb = sub(2500)
    Sn, Sb = sub(400)
    Tn, Tb = sub(60)
    mn, mx = mc.compute_aabb(P, Bi, B, True)
    specs = [  # (points, ids, centres, ids, radius, use_pdf, mask)
        (Tn, Tb, Tn, Tb, 0.6, 1, 7),      # tiny: tr_small
        (Mn, Mb, Sn, Sb, 0.25, 1, 7),     # 2 500 rows transposed, list of ~100 k edges: chain transposition under a small plan
        (Mn, Mb, Mn, Mb, 0.12, 1, 7),     # same level
        (P, Bi, P, Bi, 0.12, 1, 7),       # large: own chains
        (Sn, Sb, Sn, Sb, 0.4, 0, 7),      # no densities
        (Mn, Mb, Sn, Sb, 0.25, 1, 2),     # transposed plan alone (the list comes with it)
        (Mn, Mb, Sn, Sb, 0.25, 1, 4),     # list alone
        (Mn, Mb, Sn, Sb, 0.25, 1, 1),     # forward plan alone
        (Sn, Sb, Tn, Tb, 0.5, 1, 6),
    ]
    specs = specs + [specs[1], specs[0], specs[2], specs[8], specs[4]]
    st = torch.cuda.current_stream().cuda_stream

    def build(batched, avg):
        keep, handles, pieces, sizes = [], [], [], []
        slots = torch.empty(len(specs), dtype=torch.int32).pin_memory()
        for k, (p, b, c, cb, r, up, mask) in enumerate(specs):
            n, m = p.shape[0], c.shape[0]
            nc = mc._num_cells(mn, mx, B, r, True)
            cap = min(900 * m, 4_000_000)
            nbytes = lib.mccnn_geometry_bytes(n, m, B, nc, cap, 1)
            buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            h = lib.mccnn_geometry_create()
            keep.append(buf)
            handles.append(h)
            vp = lambda t: C.c_void_p(t.data_ptr())
            check(lib.mccnn_geometry_build(C.c_void_p(h), vp(p), vp(b), n, vp(c), vp(cb), m, vp(mn), vp(mx), B, nc, C.c_float(r), 1,
                                           C.c_float(0.25), up, cap, None, C.c_void_p(buf.data_ptr()), C.c_size_t(nbytes),
                                           C.c_void_p(slots.data_ptr() + 4 * k), C.c_void_p(st)), "geometry_build")
        torch.cuda.synchronize()
        wsmax = 256
        for k, (p, b, c, cb, r, up, mask) in enumerate(specs):
            e = int(slots[k])
            assert 0 < e <= min(900 * c.shape[0], 4_000_000), (k, e)
            sizes.append(e)
            mine = {}
            for bit in (1, 2, 4, 8):
                if not ((mask | 8) & bit):
                    continue
                nb, wb = C.c_longlong(0), C.c_longlong(0)
                check(lib.mccnn_geometry_piece_bytes(C.c_void_p(handles[k]), bit, C.byref(nb), C.byref(wb)), "piece_bytes")
                wsmax = max(wsmax, wb.value)
                if bit == 2 and not (mask & 4):   # a transposed plan needs the list
                    nl = C.c_longlong(0)
                    check(lib.mccnn_geometry_piece_bytes(C.c_void_p(handles[k]), 4, C.byref(nl), C.byref(wb)), "piece_bytes")
                    wsmax = max(wsmax, wb.value)
                    t = torch.zeros(max(nl.value, 256), dtype=torch.uint8, device="cuda")
                    check(lib.mccnn_geometry_attach(C.c_void_p(handles[k]), 4, C.c_void_p(t.data_ptr()), C.c_size_t(t.numel())), "attach")
                    mine[4] = t
                if nb.value <= 0:
                    continue
                t = torch.zeros(nb.value, dtype=torch.uint8, device="cuda")
                check(lib.mccnn_geometry_attach(C.c_void_p(handles[k]), bit, C.c_void_p(t.data_ptr()), C.c_size_t(nb.value)), "attach")
                mine[bit] = t
            pieces.append(mine)
        if batched:
            hs = (C.c_void_p * len(specs))(*handles)
            wh = (C.c_int * len(specs))(*[s[6] for s in specs])
            wsb = lib.mccnn_geometry_prebuild_batch_ws_bytes(hs, wh, len(specs))
            assert wsb > 0
            ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
            check(lib.mccnn_geometry_prebuild_batch(hs, wh, len(specs), avg, C.c_void_p(ws.data_ptr()), C.c_size_t(wsb), C.c_void_p(st)),
                  "geometry_prebuild_batch")
        else:
            ws = torch.empty(wsmax, dtype=torch.uint8, device="cuda")
            for k, s in enumerate(specs):
                check(lib.mccnn_geometry_prebuild(C.c_void_p(handles[k]), s[6], avg, C.c_void_p(ws.data_ptr()), C.c_size_t(wsmax),
                                                  C.c_void_p(st)), "geometry_prebuild")
        torch.cuda.synchronize()
        out = [{bit: t.cpu() for bit, t in mine.items() if bit != 8} for mine in pieces]
        for h in handles:
            lib.mccnn_geometry_destroy(C.c_void_p(h))
        return sizes, out

    for avg in (0, 1):
        (ea, a), (eb, b_) = build(False, avg), build(True, avg)
        assert ea == eb
        chain = 0
        for k, (x, y) in enumerate(zip(a, b_)):
            assert x.keys() == y.keys()
            for bit in x:
                assert torch.equal(x[bit], y[bit]), (avg, k, bit, ea[k])
                assert x[bit].any(), (k, bit)
        assert ea[0] <= 4096 and ea[1] > 16384 and ea[3] > 262144, ea   # the three regimes were met


@pytest.mark.gpu
def test_batch_entries_error_behaviour(mc):
    """Error codes of the two batch entries, as the single entries give them: no requests is not an error, a null array is
    MCCNN_E_BADARG, a request with a too small buffer fails as mccnn_geometry_build does and NOTHING of the batch is
    launched (every request is checked before the first launch); prebuild_batch with an unbuilt geometry: BADARG."""
    import ctypes as C
    import torch
    from mccnn_amd import _lib
    lib = _lib.load()
    lib.mccnn_geometry_create.restype = C.c_void_p
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.mccnn_geometry_build_batch(None, 0, st) != 0                   # MCCNN_E_BADARG

    class Req(C.Structure):
        _fields_ = [("geometry", C.c_void_p), ("pts", C.c_void_p), ("batch_ids", C.c_void_p), ("n", C.c_int),
                    ("centres", C.c_void_p), ("centre_batch_ids", C.c_void_p), ("m", C.c_int), ("aabb_min", C.c_void_p),
                    ("aabb_max", C.c_void_p), ("batch_size", C.c_int), ("num_cells", C.c_int), ("radius", C.c_float),
                    ("scale_inv", C.c_int), ("window", C.c_float), ("use_pdf", C.c_int), ("e_capacity", C.c_int),
                    ("grid_from", C.c_void_p), ("buffer", C.c_void_p), ("buffer_bytes", C.c_size_t), ("total_host", C.c_void_p)]
    reqs = (Req * 2)()
    assert lib.mccnn_geometry_build_batch(C.byref(reqs), 0, st) == 0          # nothing to do
    pts, bids = make_cloud(2000, 2, 3, "uniform", True)
    P, Bi = _t(pts), _t(bids)
    mn, mx = mc.compute_aabb(P, Bi, 2, True)
    nc = mc._num_cells(mn, mx, 2, 0.1, True)
    n = P.shape[0]
    cap = 64 * n
    nbytes = lib.mccnn_geometry_bytes(n, n, 2, nc, cap, 1)
    bufs = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(2)]
    slots = torch.zeros(2, dtype=torch.int32).pin_memory()
    hs = [lib.mccnn_geometry_create() for _ in range(2)]
    for k in range(2):
        reqs[k] = Req(hs[k], P.data_ptr(), Bi.data_ptr(), n, P.data_ptr(), Bi.data_ptr(), n, mn.data_ptr(), mx.data_ptr(), 2, nc, 0.1, 1,
                      0.25, 1, cap, None, bufs[k].data_ptr(), nbytes if k == 0 else 1024, slots.data_ptr() + 4 * k)
    l0 = lib.mccnn_debug_launch_count()
    rc = lib.mccnn_geometry_build_batch(C.byref(reqs), 2, st)
    assert rc != 0 and lib.mccnn_debug_launch_count() == l0                   # second request's buffer too small: nothing launched
    # prebuild of geometries that were never built
    arr = (C.c_void_p * 2)(*hs)
    what = (C.c_int * 2)(7, 7)
    ws = torch.empty(4096, dtype=torch.uint8, device="cuda")
    lib.mccnn_geometry_prebuild_batch_ws_bytes.restype = C.c_size_t
    assert lib.mccnn_geometry_prebuild_batch_ws_bytes(arr, what, 2) == 0
    assert lib.mccnn_geometry_prebuild_batch(arr, what, 2, 0, C.c_void_p(ws.data_ptr()), C.c_size_t(4096), st) != 0
    # ... and the repaired batch builds
    reqs[1].buffer_bytes = nbytes
    assert lib.mccnn_geometry_build_batch(C.byref(reqs), 2, st) == 0
    torch.cuda.synchronize()
    assert int(slots[0]) == int(slots[1]) > 0
    for h in hs:
        lib.mccnn_geometry_destroy(C.c_void_p(h))
