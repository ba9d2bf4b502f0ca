"""all levels of a PointHierarchy built with ONE host read-back (device-side point counts,
MCConvModule.point_hierarchy_levels) against the op-by-op chain of MCConvBuilder.py:101-128 and against the oracle:
identical points, batch ids, sample indices and features at every level, identical feature gradients."""
import math

import numpy as np
import pytest

from tests.helpers import make_cloud, make_room

pytestmark = pytest.mark.gpu


def _levels(ph):
    return [(p.detach().cpu().numpy(), b.cpu().numpy(), f.detach().cpu().numpy()) for p, b, f in
            zip(ph.points_, ph.batchIds_, ph.features_)], [i.cpu().numpy() for i in ph.sampledIndexs_]


@pytest.mark.parametrize("case", ["relative_batched", "room_absolute"])
def test_fused_hierarchy_equals_the_op_chain(mc, oracle, case):
    import torch
    import mccnn_amd.MCConvBuilder as MB
    if case == "relative_batched":
        pts, bids = make_cloud(3000, 5, 3, "clustered", True)
        B, radii, rel = 5, [0.1, 0.4, math.sqrt(3.0) + 0.1], True
    else:
        pts = make_room(100000, 20180601)
        bids = np.zeros((len(pts), 1), np.int32)
        B, radii, rel = 1, [0.1, 0.2, 0.4, 0.8], False
    feats = np.random.default_rng(1).random((len(pts), 3), dtype=np.float32)
    P, Bi = torch.from_numpy(pts).cuda(), torch.from_numpy(bids).cuda()
    res = {}
    for fused in (True, False):
        MB.FUSED_HIERARCHY = fused
        try:
            F = torch.from_numpy(feats).cuda().requires_grad_(True)
            ph = MB.PointHierarchy(P, F, Bi, radii, "PH", B, rel)
            (ph.features_[-1] * torch.linspace(1, 2, 3, device="cuda")).sum().backward()
            res[fused] = (_levels(ph), F.grad.cpu().numpy())
        finally:
            MB.FUSED_HIERARCHY = True
    (lv_f, idx_f), g_f = res[True]
    (lv_o, idx_o), g_o = res[False]
    assert len(lv_f) == len(radii) + 1
    for (pf, bf, ff), (po, bo, fo) in zip(lv_f, lv_o):
        assert np.array_equal(pf, po) and np.array_equal(bf, bo) and np.array_equal(ff, fo)
    for a, b in zip(idx_f, idx_o):
        assert np.array_equal(a, b)
    assert np.array_equal(g_f, g_o)
    # and against the oracle, level by level (hierarchy of the relative case only: the sequential Poisson oracle needs
    # minutes for the 100k room)
    if case == "relative_batched":
        mn, mx = oracle.compute_aabb(pts, bids, B, rel)
        cp, cb, cf = pts, bids, feats
        for l, r in enumerate(radii):
            k, i = oracle.sort_points_step1(cp, cb, mn, mx, B, r, rel)
            sp, sb, sf, cl = oracle.sort_points_step2(cp, cb, cf, k, i, mn, mx, B, r, rel)
            op, ob, oi = oracle.poisson_sampling(sp, sb, cl, mn, mx, r, B, rel)
            of = oracle.get_sampled_features(oi, sf)
            ti = oracle.transform_indexs(oi, i)
            assert np.array_equal(lv_f[l + 1][0], op) and np.array_equal(lv_f[l + 1][1], ob)
            assert np.array_equal(lv_f[l + 1][2], of) and np.array_equal(idx_f[l], ti)
            cp, cb, cf = op, ob, of


@pytest.mark.parametrize("dataflow", [False, 2], ids=["dataflow_off", "dataflow_gives_up"])
def test_fused_hierarchy_honours_the_poisson_switch_and_falls_back(mc, dataflow):
    """MCConvModule.POISSON_DATAFLOW also governs the fused hierarchy: False = no single-launch Poisson kernel anywhere
    (the fused path declines, the op-by-op chain runs the phased form); 2 = the single-launch kernel gives up at the
    first unfinished dependency, a level reports -1, point_hierarchy_levels() returns None and PointHierarchy repeats
    op by op. Either way the hierarchy equals the default one."""
    import torch
    import mccnn_amd.MCConvBuilder as MB
    pts, bids = make_cloud(3000, 4, 5, "clustered", True)
    B, radii = 4, [0.1, 0.4]
    P, Bi = torch.from_numpy(pts).cuda(), torch.from_numpy(bids).cuda()
    F = torch.from_numpy(np.random.default_rng(2).random((len(pts), 2), dtype=np.float32)).cuda()
    ref = MB.PointHierarchy(P, F, Bi, radii, "PH", B, True)
    mn, mx = ref.aabbMin_, ref.aabbMax_
    before = mc.POISSON_FALLBACKS
    mc.POISSON_DATAFLOW = dataflow
    try:
        fused = mc.point_hierarchy_levels(P, Bi, mn, mx, radii, B, True)
        assert fused is None                      # declined (False) or a level timed out (2)
        ph = MB.PointHierarchy(P, F, Bi, radii, "PH", B, True)
    finally:
        mc.POISSON_DATAFLOW = True
    if dataflow == 2:
        assert mc.POISSON_FALLBACKS > before       # the op-by-op chain had to repeat with the phased form, too
    assert len(ph.points_) == len(ref.points_) == 3
    for a, b in zip(ph.points_ + ph.batchIds_ + ph.features_ + ph.sampledIndexs_,
                    ref.points_ + ref.batchIds_ + ref.features_ + ref.sampledIndexs_):
        assert torch.equal(a, b)
    assert mc.point_hierarchy_levels(P[:0], Bi[:0], mn, mx, radii, B, True) is None   # empty cloud: op-by-op chain
    assert mc.point_hierarchy_levels(P, Bi, mn, mx, [], B, True) == []


@pytest.mark.parametrize("case", ["relative_batched", "room_absolute"])
def test_prefetched_hierarchy_equals_the_inline_one(mc, case):
    """PointHierarchy.prefetch(): boxes and levels built on a stream of their own by the extension's helper thread, under
    other work of the calling stream -- same tensors as the inline build, bit for bit; several requests in flight; a
    request nobody adopts; a handle offered to the wrong inputs."""
    import torch
    import mccnn_amd.MCConvBuilder as MB
    from mccnn_amd import native
    from mccnn_amd.MCConvModule import InvalidArgumentError
    if not native.side_streams_available():
        pytest.skip("PointHierarchy.prefetch() needs the torch extension (it returns None without it: inline build)")
    if case == "relative_batched":
        pts, bids = make_cloud(3000, 5, 3, "clustered", True)
        B, radii, rel = 5, [0.1, 0.4, math.sqrt(3.0) + 0.1], True
    else:
        pts = make_room(100000, 20180601)
        bids = np.zeros((len(pts), 1), np.int32)
        B, radii, rel = 1, [0.1, 0.2, 0.4, 0.8], False
    feats = np.random.default_rng(1).random((len(pts), 3), dtype=np.float32)
    P, Bi = torch.from_numpy(pts).cuda(), torch.from_numpy(bids).cuda()
    F = torch.from_numpy(feats).cuda().requires_grad_(True)
    ref = MB.PointHierarchy(P, F, Bi, radii, "PH", B, rel)
    (ref.features_[-1] * torch.linspace(1, 2, 3, device="cuda")).sum().backward()
    g_ref = F.grad.clone()
    (lv_r, idx_r) = _levels(ref)
    busy = torch.randn(2048, 2048, device="cuda")
    up = torch.cuda.Stream()
    for rep in range(3):
        # what the build waits for: the calling stream (default) | nothing: the inputs are complete | an event of the
        # stream that produced them (here: a copy of the points on a stream of its own)
        if rep == 0:
            after, Pin, Bin = None, P, Bi
        elif rep == 1:
            torch.cuda.synchronize()
            after, Pin, Bin = True, P, Bi
        else:
            torch.cuda.synchronize()
            with torch.cuda.stream(up):
                Pin, Bin = P.clone(), Bi.clone()
                after = torch.cuda.Event()
                after.record()
        h = MB.PointHierarchy.prefetch(Pin, Bin, radii, B, rel, after=after)
        assert h is not None
        h2 = MB.PointHierarchy.prefetch(P, Bi, radii, B, rel)       # a second request in flight, never adopted
        for _ in range(4):
            busy = torch.tanh(busy @ busy * 1e-3)                     # work of the calling stream the build runs under
        F.grad = None
        torch.cuda.current_stream().wait_stream(up)
        ph = MB.PointHierarchy(Pin, F, Bin, radii, "PH", B, rel, prefetched=h)
        del h2
        (ph.features_[-1] * torch.linspace(1, 2, 3, device="cuda")).sum().backward()
        (lv, idx) = _levels(ph)
        assert len(lv) == len(radii) + 1
        for (pf, bf, ff), (po, bo, fo) in zip(lv, lv_r):
            assert np.array_equal(pf, po) and np.array_equal(bf, bo) and np.array_equal(ff, fo)
        for a, b in zip(idx, idx_r):
            assert np.array_equal(a, b)
        assert torch.equal(ph.aabbMin_, ref.aabbMin_) and torch.equal(ph.aabbMax_, ref.aabbMax_)
        assert torch.equal(F.grad, g_ref)
    # the adopted hierarchy serves a convolution builder like any other (cell counts of absolute radii come from the
    # extent the helper thread read back)
    builder = MB.ConvolutionBuilder(KDEWindow=0.25, relativeRadius=rel)
    ref_b = MB.ConvolutionBuilder(KDEWindow=0.25, relativeRadius=rel)
    f1 = torch.ones((len(pts), 1), device="cuda")
    o1 = builder.create_convolution("c", ph, 0, f1, 1, radii[0] * (1.0 if rel else 1.0), ph, 1, True, 8)
    ref_b.load_state_dict(builder.state_dict(), strict=False)
    o2 = ref_b.create_convolution("c", ref, 0, f1, 1, radii[0], ref, 1, True, 8)
    assert torch.equal(o1, o2)
    # a handle is bound to its request
    h = MB.PointHierarchy.prefetch(P, Bi, radii, B, rel)
    with pytest.raises(InvalidArgumentError):
        MB.PointHierarchy(P, F, Bi, radii[:-1], "PH", B, rel, prefetched=h)
    P2 = P.clone()
    with pytest.raises(InvalidArgumentError):
        MB.PointHierarchy(P2, F, Bi, radii, "PH", B, rel, prefetched=h)
    del h
    with pytest.raises(InvalidArgumentError):
        MB.PointHierarchy.prefetch(P, Bi, radii, B, rel, after="now")
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_prefetched_hierarchy_gathers_the_feature_rows(mc, dtype):
    """PointHierarchy.prefetch(..., features=F): the feature rows of every level (GetSampledFeatures,
    MCConvBuilder.py:112-116) are gathered with the hierarchy on its own stream. The constructor takes them when it is
    handed the very same, unmodified tensor without a gradient -- bit-identical to the inline gathers -- and gathers
    itself (differentiably) when the rows carry a gradient, are another tensor, or were modified in between."""
    import torch
    import mccnn_amd.MCConvBuilder as MB
    from mccnn_amd import native
    if not native.side_streams_available():
        pytest.skip("PointHierarchy.prefetch() needs the torch extension")
    pts, bids = make_cloud(2500, 4, 9, "clustered", True)
    B, radii = 4, [0.1, 0.3, 0.9]
    P, Bi = torch.from_numpy(pts).cuda(), torch.from_numpy(bids).cuda()
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    F = torch.from_numpy(np.random.default_rng(4).random((len(pts), 6), dtype=np.float32)).cuda().to(td)
    ref = MB.PointHierarchy(P, F, Bi, radii, "PH", B, True)
    launches = mc._lib.load().mccnn_debug_launch_count

    def same(ph):
        assert len(ph.features_) == len(ref.features_) == 4
        for a, b in zip(ph.points_ + ph.batchIds_ + ph.features_ + ph.sampledIndexs_,
                        ref.points_ + ref.batchIds_ + ref.features_ + ref.sampledIndexs_):
            assert a.dtype == b.dtype and torch.equal(a, b)

    # 1. the very tensor: no gather launch on the calling thread at adoption
    h = MB.PointHierarchy.prefetch(P, Bi, radii, B, True, features=F)
    assert h is not None
    h.future.result()                      # (the helper thread is done: whatever is launched below is the constructor's)
    l0 = launches()
    ph = MB.PointHierarchy(P, F, Bi, radii, "PH", B, True, prefetched=h)
    assert launches() == l0
    torch.cuda.synchronize()
    same(ph)
    # 2. rows with a gradient: gathered by the constructor, and the gradient flows through every level
    Fg = F.float().clone().requires_grad_(True)
    h = MB.PointHierarchy.prefetch(P, Bi, radii, B, True, features=Fg)
    ph = MB.PointHierarchy(P, Fg, Bi, radii, "PH", B, True, prefetched=h)
    assert ph.features_[3].requires_grad
    ph.features_[3].sum().backward()
    assert float(Fg.grad.sum()) == ph.features_[3].numel()
    # 3. another tensor / a tensor modified since the request: gathered by the constructor from what it is handed
    h = MB.PointHierarchy.prefetch(P, Bi, radii, B, True, features=F)
    F2 = (F.float() * 2).to(td)
    ph = MB.PointHierarchy(P, F2, Bi, radii, "PH", B, True, prefetched=h)
    assert torch.equal(ph.features_[1].float(), ref.features_[1].float() * 2)
    Fm = F.clone()
    h = MB.PointHierarchy.prefetch(P, Bi, radii, B, True, features=Fm)
    h.future.result()
    Fm.mul_(0)
    ph = MB.PointHierarchy(P, Fm, Bi, radii, "PH", B, True, prefetched=h)
    torch.cuda.synchronize()
    assert float(ph.features_[2].float().abs().sum()) == 0.0
