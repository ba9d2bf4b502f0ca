"""The host-floor extensions against the op-by-op forms they replace (bit for bit on integers and copies):
single-workgroup kernels for tiny grids / neighbour lists, mccnn_build_grid, the one-call row-plan build,
spatial_conv(sortIndex=) and the builder's fused path."""
import numpy as np
import pytest

from tests.helpers import make_cloud, make_mlp

pytestmark = pytest.mark.gpu


@pytest.fixture
def mc():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mccnn_amd import MCConvModule
    return MCConvModule


def _t(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _chain(mc, P, Bi, F, B, radius):
    mn, mx = mc.compute_aabb(P, Bi, B, True)
    keys, idx = mc.sort_points_step1(P, Bi, mn, mx, B, radius, True)
    sP, sB, sF, cells = mc.sort_points_step2(P, Bi, F, keys, idx, mn, mx, B, radius, True)
    start, packed = mc.find_neighbors(P, Bi, sP, cells, mn, mx, radius, B, True)
    st, pt, _ = mc._transposed_neighbors(packed, sP.shape[0])
    return dict(mn=mn, mx=mx, keys=keys, idx=idx, sP=sP, sB=sB, sF=sF, cells=cells, start=start, packed=packed,
                start_t=st.clone(), perm_t=pt[:packed.shape[0]].clone())


@pytest.mark.parametrize("n_per,B,radius", [(37, 3, 0.4), (600, 3, 0.2), (2000, 1, 0.08), (5, 1, 0.9)])
def test_single_workgroup_kernels_equal_the_multi_launch_ones(mc, n_per, B, radius):
    """Tiny grids / lists take one-workgroup kernels (grid_small_step1/2, tr_small); mccnn_debug_small_kernels(0) sends the
    same input through the kernels of the large problems: every output identical."""
    import torch
    from mccnn_amd import _lib
    lib = _lib.load()
    pts, bids = make_cloud(n_per, B, 11, "clustered", True)
    rng = np.random.default_rng(3)
    P, Bi, F = _t(pts), _t(bids), _t(rng.random((len(pts), 3), dtype=np.float32))
    prev = lib.mccnn_debug_small_kernels(1)
    try:
        a = _chain(mc, P, Bi, F, B, radius)
        lib.mccnn_debug_small_kernels(0)
        b = _chain(mc, P, Bi, F, B, radius)
    finally:
        lib.mccnn_debug_small_kernels(prev)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_build_grid_equals_the_two_sort_ops(mc):
    import torch
    for n_per, B, radius, rel in ((900, 4, 0.15, True), (30000, 2, 0.05, True), (1500, 3, 0.2, False)):
        pts, bids = make_cloud(n_per, B, 5, "uniform", True)
        P, Bi = _t(pts), _t(bids)
        F = torch.ones((len(pts), 1), device="cuda")
        mn, mx = mc.compute_aabb(P, Bi, B, rel)
        keys, idx = mc.sort_points_step1(P, Bi, mn, mx, B, radius, rel)
        sP, sB, _, cells = mc.sort_points_step2(P, Bi, F, keys, idx, mn, mx, B, radius, rel)
        gP, gB, gC, gI, gInv = mc.build_grid(P, Bi, mn, mx, B, radius, rel)
        assert torch.equal(gP, sP) and torch.equal(gB, sB) and torch.equal(gC, cells) and torch.equal(gI, idx)
        assert torch.equal(gInv[gI.long()].cpu(), torch.arange(len(pts), dtype=torch.int32))  # the inverse permutation


@pytest.mark.parametrize("fin,fout,combin", [(1, 16, True), (3, 8, True), (16, 16, False), (64, 64, False)])
def test_spatial_conv_sort_index_equals_sort_features(mc, fin, fout, combin):
    """spatial_conv(sortIndex=) == spatial_conv(sort_features(...)): outputs and all seven gradients, bit for bit."""
    import torch
    pts, bids = make_cloud(1500, 2, 21, "uniform", True)
    rng = np.random.default_rng(9)
    P, Bi = _t(pts), _t(bids)
    B, radius = 2, 0.15
    mn, mx = mc.compute_aabb(P, Bi, B, True)
    sP, sB, cells, idx, inv = mc.build_grid(P, Bi, mn, mx, B, radius, True)
    start, packed = mc.find_neighbors(P, Bi, sP, cells, mn, mx, radius, B, True)
    pdfs = mc.compute_pdf(sP, sB, mn, mx, start, packed, 0.2, radius, B, True)
    nb = (fin * fout + 7) // 8 if combin else (fin + 7) // 8
    w = make_mlp(nb, 4)
    outF = fout if combin else fin
    og = _t(rng.random((len(pts), outF), dtype=np.float32))
    res = []
    for fused in (False, True, "in place"):
        F = _t(rng.random((len(pts), fin), dtype=np.float32) if not res else res[0][2]).requires_grad_(True)
        ws = [_t(w[k]).requires_grad_(True) for k in ("w1", "w2", "w3", "b1", "b2", "b3")]
        if fused == "in place":  # depth-wise layers: the row kernels read the unsorted rows where they lie (featIndex)
            out = mc.spatial_conv(sP, F, sB, pdfs, P, start, packed, mn, mx, *ws, fout, combin, B, radius, True, True,
                                  sortIndex=idx, featIndex=inv)
        elif fused:
            out = mc.spatial_conv(sP, F, sB, pdfs, P, start, packed, mn, mx, *ws, fout, combin, B, radius, True, True,
                                  sortIndex=idx)
        else:
            out = mc.spatial_conv(sP, mc.sort_features(F, idx), sB, pdfs, P, start, packed, mn, mx, *ws, fout, combin, B,
                                  radius, True, True)
        grads = torch.autograd.grad([out], [F] + ws, [og])
        res.append((out.detach(), grads, F.detach().cpu().numpy()))
    deterministic = not combin  # combin layers add their per-edge feature gradients with float atomics
    for other in res[1:]:
        assert torch.equal(res[0][0], other[0])
        for ga, gb in zip(res[0][1], other[1]):
            if deterministic:
                assert torch.equal(ga, gb)
            else:
                assert float((ga - gb).abs().max()) <= 1e-5 * float(ga.abs().max())


@pytest.mark.parametrize("n_per,B,radius,fout,scale_inv,avg,pool", [
    (1500, 2, 0.15, 16, True, True, False),     # ragged rows, relative radius, averaged
    (4000, 1, 0.08, 64, False, False, False),   # absolute radius, plain sums, the headline layer's shape
    (900, 3, 0.3, 8, True, True, True),         # pooling: centres are a subset of the points plus centres with EMPTY rows
    (30, 1, 0.5, 16, True, False, False),       # less than one iteration of one wave
])
def test_f1_forward_four_edges_per_lane_equals_the_chunk_kernel(mc, n_per, B, radius, fout, scale_inv, avg, pool):
    """Combin layers with one input feature: the forward edge pass with four edges per lane (lists of >= 2 000 000 edges by
    default) against the 64-edge-chunk kernel on the same inputs -- outputs and all seven gradients within float
    summation order (the backward pass consumes the per-edge records the forward pass wrote)."""
    import torch
    from mccnn_amd import _lib
    lib = _lib.load()
    pts, bids = make_cloud(n_per, B, 33, "clustered", True)
    rng = np.random.default_rng(10)
    P, Bi = _t(pts), _t(bids)
    mn, mx = mc.compute_aabb(P, Bi, B, scale_inv)
    sP, sB, cells, idx, inv = mc.build_grid(P, Bi, mn, mx, B, radius, scale_inv)
    if pool:
        sel = np.sort(rng.choice(len(pts), len(pts) // 7, replace=False))
        far = np.full((5, 3), 40.0, dtype=np.float32) + rng.random((5, 3), dtype=np.float32)  # no neighbours at all
        C = _t(np.concatenate([pts[sel][:50], far[:2], pts[sel][50:], far[2:]]))
        Cb = _t(np.concatenate([bids[sel][:50], bids[:2] * 0, bids[sel][50:], bids[:3] * 0]))
    else:
        C, Cb = P, Bi
    start, packed = mc.find_neighbors(C, Cb, sP, cells, mn, mx, radius, B, scale_inv)
    pdfs = mc.compute_pdf(sP, sB, mn, mx, start, packed, 0.2, radius, B, scale_inv)
    nb = (fout + 7) // 8
    w = make_mlp(nb, 6)
    og = _t(rng.random((C.shape[0], fout), dtype=np.float32))
    f_np = rng.random((len(pts), 1), dtype=np.float32) - 0.3
    res = []
    prev = lib.mccnn_debug_f1_x4_min_edges(2 ** 31 - 1)
    try:
        for min_edges in (2 ** 31 - 1, 0):
            lib.mccnn_debug_f1_x4_min_edges(min_edges)
            F = _t(f_np).requires_grad_(True)
            ws = [_t(w[k]).requires_grad_(True) for k in ("w1", "w2", "w3", "b1", "b2", "b3")]
            out = mc.spatial_conv(sP, F, sB, pdfs, C, start, packed, mn, mx, *ws, fout, True, B, radius, scale_inv, avg)
            grads = torch.autograd.grad([out], [F] + ws, [og])
            res.append((out.detach(), grads))
    finally:
        lib.mccnn_debug_f1_x4_min_edges(prev)
    ref, got = res
    scale = float(ref[0].abs().max())
    assert float((ref[0] - got[0]).abs().max()) <= 2e-5 * scale
    if pool:  # rows without neighbours: exact zeros
        st = torch.cat([start.flatten().long(), torch.tensor([packed.shape[0]], device=start.device)])
        empty = (st[1:] == st[:-1]).nonzero().flatten()
        assert empty.numel() >= 5 and float(got[0][empty].abs().max()) == 0.0
    for ga, gb in zip(ref[1], got[1]):
        assert float((ga - gb).abs().max()) <= 2e-5 * max(float(ga.abs().max()), 1e-20)


def test_builder_fused_path_equals_the_op_chain(mc):
    """ConvolutionBuilder(fuseSort=True) (build_grid + search / KDE back to back + sortIndex) against fuseSort=False (the
    reference's op sequence): same outputs, same gradients, over a two-level hierarchy with shared grids."""
    import torch
    from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder
    pts, bids = make_cloud(3000, 3, 77, "clustered", True)
    rng = np.random.default_rng(1)
    P, Bi = _t(pts), _t(bids)
    F0 = _t(rng.random((len(pts), 1), dtype=np.float32))
    outs = []
    for fuse in (False, True):
        torch.manual_seed(3)
        ph = PointHierarchy(P, F0, Bi, [0.1], "PH", 3, True)
        b = ConvolutionBuilder(KDEWindow=0.25, relativeRadius=True, fuseSort=fuse)
        n1 = ph.points_[1].shape[0]
        f_a = F0.clone().requires_grad_(True)
        f_b = _t(np.random.default_rng(2).random((len(pts), 16), dtype=np.float32)).requires_grad_(True)
        f_c = _t(np.random.default_rng(3).random((n1, 32), dtype=np.float32)).requires_grad_(True)
        for rep in range(2):  # the second pass has edge-count guesses: the deferred search + KDE path
            b.reset()
            o1 = b.create_convolution("C1", ph, 0, f_a, 1, 0.1, outNumFeatures=16, multiFeatureConv=True)
            o2 = b.create_convolution("C2", ph, 0, f_b, 16, 0.1)                 # cached grid, cached list
            o3 = b.create_convolution("P1", ph, 0, f_b, 16, 0.2, ph, 1)          # pooling
            o4 = b.create_convolution("C3", ph, 1, f_c, 32, 0.4)
        loss_in = [o1, o2, o3, o4]
        ogs = [torch.ones_like(o) for o in loss_in]
        grads = torch.autograd.grad(loss_in, [f_a, f_b, f_c] + list(b.parameters()), ogs)
        outs.append(([o.detach() for o in loss_in], grads))
    for x, y in zip(outs[0][0], outs[1][0]):
        assert torch.equal(x, y)
    for x, y in zip(outs[0][1], outs[1][1]):
        assert float((x - y).abs().max()) <= 1e-6 * max(float(x.abs().max()), 1e-30)
