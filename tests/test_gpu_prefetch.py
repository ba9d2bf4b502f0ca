"""ConvolutionBuilder.prefetch_geometry: geometry of the next batch computed on a side stream and installed by reset()
gives the same convolution as the inline path (integer outputs bit-exact, floats within the feature tolerance)."""
import numpy as np
import pytest

from tests.helpers import make_cloud

pytestmark = pytest.mark.gpu

# Two prefetch protocols: "native" -- the geometry is one buffer of the native step executor, allocated on the caller's
# stream and written on a side stream forked behind it (ConvolutionBuilder.__prefetch_native__, the default) -- and "ops" --
# the op-by-op geometry on the builder's side stream with its tensor-lifetime bookkeeping (ConvolutionBuilder(native=False)).
PROTOCOLS = ["native", "ops"]


def _builder(protocol, **kw):
    from mccnn_amd.MCConvBuilder import ConvolutionBuilder
    from mccnn_amd import native
    if protocol == "native" and not native.side_streams_available():
        pytest.skip("torch extension not built")
    return ConvolutionBuilder(native=(protocol == "native"), **kw)


def _parked(builder):
    return builder.prefetched_ is not None or bool(builder.prefetchedGeo_)


@pytest.mark.parametrize("protocol", PROTOCOLS)
def test_prefetched_geometry_equals_inline(mc, protocol):
    import torch
    from mccnn_amd.MCConvBuilder import PointHierarchy
    pts, bids = make_cloud(3000, 3, 23, "clustered", True)
    rng = np.random.default_rng(5)
    P = torch.from_numpy(pts).cuda()
    Bi = torch.from_numpy(bids).cuda()
    F = torch.from_numpy(rng.random((len(pts), 1), dtype=np.float32)).cuda().requires_grad_(True)
    og = torch.from_numpy(rng.random((len(pts), 16), dtype=np.float32)).cuda()
    ph = PointHierarchy(P, F, Bi, [], "PH", 3, True)
    torch.manual_seed(3)
    builder = _builder(protocol, KDEWindow=0.2, relativeRadius=True)

    def run():
        F.grad = None
        for p in builder.parameters():
            p.grad = None
        out = builder.create_convolution("Conv", ph, 0, F, 1, 0.15, outNumFeatures=16, multiFeatureConv=True)
        out.backward(og)
        neigh = next(iter(builder.cacheNeighs_.values()))
        return (out.detach().cpu().numpy(), F.grad.cpu().numpy(), [p.grad.cpu().numpy() for p in builder.parameters()],
                neigh[0].cpu().numpy(), neigh[1].cpu().numpy())

    builder.reset()
    ref = run()                                   # inline geometry
    for _ in range(3):                            # three pipelined steps: prefetch under the convolution, install, use
        builder.prefetch_geometry(ph, 0, 0.15)
        assert _parked(builder)
        builder.reset()
        assert not _parked(builder) and len(builder.cacheNeighs_) == 1 and len(builder.cachePDFs_) == 1
        if protocol == "native":
            assert next(iter(builder.cacheGeo_.values())).core.side >= 0     # built on a side stream
        got = run()
        assert np.array_equal(got[3], ref[3]) and np.array_equal(got[4], ref[4])      # start indices, packed neighbours
        assert np.array_equal(got[0], ref[0])                                          # forward: deterministic
        scale = np.abs(ref[1]).max()
        assert np.abs(got[1] - ref[1]).max() <= 1e-5 * scale                          # float atomics: order varies
        for g, r in zip(got[2], ref[2]):
            assert np.abs(g - r).max() <= 1e-5 * max(np.abs(r).max(), 1e-30)
    torch.cuda.synchronize()


@pytest.mark.parametrize("protocol", PROTOCOLS)
def test_prefetched_transposed_list_depthwise(mc, protocol):
    """Depth-wise layer: the transposed neighbour list its backward needs is built on the side stream at reset();
    the gradients equal the inline path's (the transposed gather is deterministic: bit for bit)."""
    import torch
    from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder
    pts, bids = make_cloud(2500, 2, 29, "uniform", True)
    rng = np.random.default_rng(6)
    P = torch.from_numpy(pts).cuda()
    Bi = torch.from_numpy(bids).cuda()
    F = torch.from_numpy(rng.random((len(pts), 16), dtype=np.float32)).cuda().requires_grad_(True)
    og = torch.from_numpy(rng.random((len(pts), 16), dtype=np.float32)).cuda()
    ph = PointHierarchy(P, F, Bi, [], "PH", 2, True)
    torch.manual_seed(4)
    builder = _builder(protocol, KDEWindow=0.2, relativeRadius=True)

    def run():
        F.grad = None
        for p in builder.parameters():
            p.grad = None
        out = builder.create_convolution("Conv", ph, 0, F, 16, 0.2, multiFeatureConv=False)
        out.backward(og)
        return out.detach().cpu().numpy(), F.grad.cpu().numpy(), [p.grad.cpu().numpy() for p in builder.parameters()]

    builder.reset()
    ref = run()
    for _ in range(2):
        builder.prefetch_geometry(ph, 0, 0.2, transposed=True)
        builder.reset()
        if protocol == "ops":
            packed = next(iter(builder.cacheNeighs_.values()))[1]
            assert getattr(packed, "_mccnn_transposed", None) is not None  # started at reset(), on the side stream
        else:   # transposed list (4) and transposed row plan (2) attached and started at reset()
            assert next(iter(builder.cacheGeo_.values())).core.have & 6 == 6
        got = run()
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
        for g, r in zip(got[2], ref[2]):
            assert np.array_equal(g, r)
    torch.cuda.synchronize()


def test_native_prefetch_with_too_small_size_guess_is_repaired(mc):
    """Native protocol: a prefetched geometry whose list outgrew the guessed capacity is built again -- inline, exact -- by
    the first layer that uses it."""
    import torch
    from mccnn_amd import native
    from mccnn_amd.MCConvBuilder import PointHierarchy
    pts, bids = make_cloud(2000, 2, 31, "uniform")
    rng = np.random.default_rng(8)
    P = torch.from_numpy(pts).cuda()
    Bi = torch.from_numpy(bids).cuda()
    F = torch.from_numpy(rng.random((len(pts), 1), dtype=np.float32)).cuda()
    ph = PointHierarchy(P, F, Bi, [], "PH", 2, True)
    torch.manual_seed(5)
    builder = _builder("native", KDEWindow=0.2, relativeRadius=True)
    builder.reset()
    ref = builder.create_convolution("Conv", ph, 0, F, 1, 0.2, outNumFeatures=8, multiFeatureConv=True).detach().cpu().numpy()
    e_ref = next(iter(builder.cacheNeighs_.values()))[1].shape[0]
    for k in list(native._EDGE_GUESS):
        native._EDGE_GUESS[k] = 16                # far below the ~1e5 edges of this cloud
    builder.prefetch_geometry(ph, 0, 0.2)
    geo = next(iter(builder.prefetchedGeo_.values()))[0]
    assert geo.e_cap == 16
    builder.reset()
    got = builder.create_convolution("Conv", ph, 0, F, 1, 0.2, outNumFeatures=8, multiFeatureConv=True).detach().cpu().numpy()
    st, pk = next(iter(builder.cacheNeighs_.values()))
    assert pk.shape[0] == e_ref and geo.e_cap >= e_ref
    assert np.array_equal(got, ref)
    torch.cuda.synchronize()


def test_prefetch_with_too_small_size_guess_is_repaired(mc):
    """The deferred search + KDE size their lists from the last total of the shape; a guess that is too small is
    detected when the total is read and both ops are repeated with the exact size."""
    import torch
    from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder
    pts, bids = make_cloud(2000, 2, 31, "uniform")
    rng = np.random.default_rng(8)
    P = torch.from_numpy(pts).cuda()
    Bi = torch.from_numpy(bids).cuda()
    F = torch.from_numpy(rng.random((len(pts), 1), dtype=np.float32)).cuda()
    ph = PointHierarchy(P, F, Bi, [], "PH", 2, True)
    torch.manual_seed(5)
    builder = ConvolutionBuilder(KDEWindow=0.2, relativeRadius=True, native=False)
    builder.reset()
    ref = builder.create_convolution("Conv", ph, 0, F, 1, 0.2, outNumFeatures=8, multiFeatureConv=True).detach().cpu().numpy()
    e_ref = next(iter(builder.cacheNeighs_.values()))[1].shape[0]
    assert len(mc._EDGE_GUESS) > 0
    for k in list(mc._EDGE_GUESS):
        mc._EDGE_GUESS[k] = 16                    # far below the ~1e5 edges of this cloud
    builder.prefetch_geometry(ph, 0, 0.2)
    h = next(iter(builder.prefetched_[1].values()))
    assert hasattr(h, "finalize")                 # the deferred path was taken
    builder.reset()
    st, pk = next(iter(builder.cacheNeighs_.values()))
    assert pk.shape[0] == e_ref and next(iter(builder.cachePDFs_.values())).shape[0] == e_ref
    got = builder.create_convolution("Conv", ph, 0, F, 1, 0.2, outNumFeatures=8, multiFeatureConv=True).detach().cpu().numpy()
    assert np.array_equal(got, ref)
    torch.cuda.synchronize()


def test_deferred_search_and_kde_equal_the_two_ops(mc):
    """find_neighbors_pdf_deferred (lists sized by a guess, KDE reading the edge count from device memory) returns the
    same start indices, neighbour list and pdfs -- bit for bit -- as find_neighbors + compute_pdf."""
    import torch
    pts, bids = make_cloud(1800, 3, 37, "clustered", True)
    P = torch.from_numpy(pts).cuda()
    Bi = torch.from_numpy(bids).cuda()
    F = torch.ones((len(pts), 1), device="cuda")
    B, r, w = 3, 0.12, 0.25
    mn, mx = mc.compute_aabb(P, Bi, B, True)
    keys, idx = mc.sort_points_step1(P, Bi, mn, mx, B, r, True)
    sP, sB, sF, cells = mc.sort_points_step2(P, Bi, F, keys, idx, mn, mx, B, r, True)
    mc.clear_caches()
    assert mc.find_neighbors_pdf_deferred(P, Bi, sP, sB, cells, mn, mx, r, B, True, w) is None   # no size guess yet
    start, packed = mc.find_neighbors(P, Bi, sP, cells, mn, mx, r, B, True)
    pdfs = mc.compute_pdf(sP, sB, mn, mx, start, packed, w, r, B, True)
    h = mc.find_neighbors_pdf_deferred(P, Bi, sP, sB, cells, mn, mx, r, B, True, w)
    assert h is not None
    st2, pk2, pdf2 = h.finalize()
    torch.cuda.synchronize()
    assert torch.equal(st2, start) and torch.equal(pk2, packed) and torch.equal(pdf2, pdfs)


@pytest.mark.parametrize("protocol", PROTOCOLS)
def test_combin_feature_gradient_through_the_transposed_list(mc, protocol):
    """Combin layer with 3 input features: with a (prefetched) transposed list the feature gradient is gathered in a
    fixed order instead of added with float atomics -- equal to the atomic form within float-sum noise, and bit-identical
    from run to run."""
    import torch
    from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder
    pts, bids = make_cloud(2500, 2, 41, "uniform", True)
    rng = np.random.default_rng(9)
    P = torch.from_numpy(pts).cuda()
    Bi = torch.from_numpy(bids).cuda()
    F = torch.from_numpy(rng.random((len(pts), 3), dtype=np.float32)).cuda().requires_grad_(True)
    og = torch.from_numpy(rng.random((len(pts), 8), dtype=np.float32)).cuda()
    ph = PointHierarchy(P, F, Bi, [], "PH", 2, True)
    torch.manual_seed(6)
    builder = _builder(protocol, KDEWindow=0.2, relativeRadius=True)

    def run():
        F.grad = None
        for p in builder.parameters():
            p.grad = None
        out = builder.create_convolution("Conv", ph, 0, F, 3, 0.2, outNumFeatures=8, multiFeatureConv=True)
        out.backward(og)
        return F.grad.cpu().numpy()

    builder.reset()
    ref = run()                                   # float atomics
    got = []
    for _ in range(2):
        builder.prefetch_geometry(ph, 0, 0.2, transposed=True)
        builder.reset()
        got.append(run())                         # transposed gather
    assert np.array_equal(got[0], got[1])
    assert np.abs(got[0] - ref).max() <= 1e-5 * np.abs(ref).max()
    torch.cuda.synchronize()


@pytest.mark.parametrize("protocol", PROTOCOLS)
def test_pipeline_over_changing_batches(mc, protocol):
    """A training loop's shape: every step convolves a DIFFERENT batch while the geometry of the next one is prefetched.
    Batch sizes and edge counts change from step to step (the deferred search sizes its lists from the last total of
    the same shape, so some guesses are too small and finalize() repairs them): every step must reproduce what the
    same batch gives without any prefetch."""
    import torch
    from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder
    rng = np.random.default_rng(9)
    batches = []
    for n_per, B, seed, kind in ((1500, 2, 31, "uniform"), (4000, 2, 32, "clustered"), (800, 2, 33, "uniform"),
                                 (3000, 2, 34, "clustered"), (4000, 2, 35, "uniform")):
        pts, bids = make_cloud(n_per, B, seed, kind, True)
        P = torch.from_numpy(pts).cuda()
        Bi = torch.from_numpy(bids).cuda()
        F = torch.from_numpy(rng.random((len(pts), 1), dtype=np.float32)).cuda().requires_grad_(True)
        og = torch.from_numpy(rng.random((len(pts), 16), dtype=np.float32)).cuda()
        batches.append((PointHierarchy(P, F, Bi, [], "PH", B, False), F, og))
    torch.manual_seed(5)
    builder = _builder(protocol, KDEWindow=0.2, relativeRadius=False)

    def conv(ph, F, og):
        F.grad = None
        for p in builder.parameters():
            p.grad = None
        out = builder.create_convolution("Conv", ph, 0, F, 1, 0.12, outNumFeatures=16, multiFeatureConv=True)
        out.backward(og)
        return out

    refs = []
    for ph, F, og in batches:                      # no prefetch: the inline path, batch by batch
        builder.reset()
        out = conv(ph, F, og)
        neigh = next(iter(builder.cacheNeighs_.values()))
        refs.append((out.detach().clone(), F.grad.clone(), neigh[1].clone()))
    order = [0, 1, 2, 3, 4, 1, 0, 4, 2, 3, 3, 0]
    builder.reset()
    builder.prefetch_geometry(batches[order[0]][0], 0, 0.12)
    for step, b in enumerate(order):
        ph, F, og = batches[b]
        builder.reset()                            # installs the geometry prefetched for THIS batch
        assert not _parked(builder) and len(builder.cacheNeighs_) == 1 and len(builder.cachePDFs_) == 1
        out = conv(ph, F, og)
        assert len(builder.cacheNeighs_) == 1      # the convolution used the installed lists, it searched nothing itself
        if step + 1 < len(order):
            builder.prefetch_geometry(batches[order[step + 1]][0], 0, 0.12)   # under the kernels just launched
        neigh = next(iter(builder.cacheNeighs_.values()))
        assert torch.equal(neigh[1], refs[b][2]), (step, b)
        assert torch.equal(out.detach(), refs[b][0]), (step, b)
        scale = float(refs[b][1].abs().max())
        assert float((F.grad - refs[b][1]).abs().max()) <= 1e-5 * scale, (step, b)
    torch.cuda.synchronize()


def test_side_stream_tensor_lifetime(mc):
    """The prefetched tensors are allocated on the side stream and read on the main one (op-by-op protocol): their memory is
    handed to the allocator with record_stream(), so a reader that comes late -- here: an autograd graph that runs its
    backward pass after the next reset() -- still finds its lists intact; plain loops and changing batches first."""
    import torch
    from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder
    pts, bids = make_cloud(2000, 2, 41, "uniform", True)
    rng = np.random.default_rng(8)
    P = torch.from_numpy(pts).cuda()
    Bi = torch.from_numpy(bids).cuda()
    F = torch.from_numpy(rng.random((len(pts), 1), dtype=np.float32)).cuda().requires_grad_(True)
    og = torch.from_numpy(rng.random((len(pts), 16), dtype=np.float32)).cuda()
    ph = PointHierarchy(P, F, Bi, [], "PH", 2, True)
    torch.manual_seed(6)
    builder = ConvolutionBuilder(KDEWindow=0.2, relativeRadius=True, native=False)   # the op-by-op protocol's bookkeeping

    def conv():
        return builder.create_convolution("Conv", ph, 0, F, 1, 0.15, outNumFeatures=16, multiFeatureConv=True)

    builder.reset()
    ref = conv()
    ref.backward(og)
    ref_grad = F.grad.clone()
    for _ in range(4):                      # the plain loop
        builder.prefetch_geometry(ph, 0, 0.15)
        builder.reset()
        F.grad = None
        out = conv()
        out.backward(og)
        assert torch.equal(out.detach(), ref.detach())
    # ... and with batches that CHANGE
    pts2, bids2 = make_cloud(1500, 2, 42, "clustered", True)
    F2 = torch.from_numpy(rng.random((len(pts2), 1), dtype=np.float32)).cuda().requires_grad_(True)
    ph2 = PointHierarchy(torch.from_numpy(pts2).cuda(), F2, torch.from_numpy(bids2).cuda(), [], "PH2", 2, True)
    og2 = torch.from_numpy(rng.random((len(pts2), 16), dtype=np.float32)).cuda()
    for i in range(6):
        nxt, Fn, ogn = (ph2, F2, og2) if i % 2 == 0 else (ph, F, og)
        builder.prefetch_geometry(nxt, 0, 0.15)
        builder.reset()
        Fn.grad = None
        o2 = builder.create_convolution("Conv", nxt, 0, Fn, 1, 0.15, outNumFeatures=16, multiFeatureConv=True)
        o2.backward(ogn)
    builder.prefetch_geometry(ph, 0, 0.15)
    builder.reset()
    F.grad = None
    late = conv()                           # graph kept alive across the next reset(): it still owns the lists
    builder.prefetch_geometry(ph, 0, 0.15)
    builder.reset()
    late.backward(og)                       # reads the retired lists AFTER the reset
    assert torch.equal(late.detach(), ref.detach())
    assert float((F.grad - ref_grad).abs().max()) <= 1e-5 * float(ref_grad.abs().max())
    torch.cuda.synchronize()


def test_native_prefetch_and_a_graph_kept_across_resets(mc):
    """Native protocol: a prefetched geometry is one buffer taken from the CALLER's stream's allocator and only written on
    the side stream; nothing about its lifetime is decided by reference counts. A graph that outlives two reset()s (and a
    prefetch of the same shape in between, whose buffers could take the memory if it were released early) still computes
    the right gradients; changing batches and a prefetch nobody uses leave nothing behind."""
    import torch
    from mccnn_amd.MCConvBuilder import PointHierarchy
    pts, bids = make_cloud(2000, 2, 41, "uniform", True)
    rng = np.random.default_rng(8)
    P = torch.from_numpy(pts).cuda()
    Bi = torch.from_numpy(bids).cuda()
    F = torch.from_numpy(rng.random((len(pts), 1), dtype=np.float32)).cuda().requires_grad_(True)
    og = torch.from_numpy(rng.random((len(pts), 16), dtype=np.float32)).cuda()
    ph = PointHierarchy(P, F, Bi, [], "PH", 2, True)
    torch.manual_seed(6)
    builder = _builder("native", KDEWindow=0.2, relativeRadius=True)

    def conv():
        return builder.create_convolution("Conv", ph, 0, F, 1, 0.15, outNumFeatures=16, multiFeatureConv=True)

    builder.reset()
    ref = conv()
    ref.backward(og)
    ref_grad = F.grad.clone()
    for _ in range(3):
        builder.reset()
        builder.prefetch_geometry(ph, 0, 0.15)      # the recommended place: before this batch's convolutions
        F.grad = None
        out = conv()
        out.backward(og)
        assert torch.equal(out.detach(), ref.detach())
    builder.reset()
    F.grad = None
    late = conv()                                   # graph kept alive ...
    for _ in range(2):                              # ... across two more steps that prefetch and convolve the same shape
        builder.prefetch_geometry(ph, 0, 0.15)
        builder.reset()
        conv().backward(og)
    F.grad = None
    late.backward(og)
    assert torch.equal(late.detach(), ref.detach())
    assert float((F.grad - ref_grad).abs().max()) <= 1e-5 * float(ref_grad.abs().max())
    builder.prefetch_geometry(ph, 0, 0.15)          # a prefetch nobody uses: dropped by the next resets
    builder.reset()
    builder.reset()
    assert not builder.cacheGeo_ and not _parked(builder)
    torch.cuda.synchronize()


@pytest.mark.parametrize("protocol", PROTOCOLS)
def test_prefetch_waits_for_a_hierarchy_built_after_reset(mc, protocol):
    """A real training loop builds the NEXT batch's PointHierarchy (host-to-device copies, compute_aabb, Poisson levels)
    on the calling stream AFTER the last reset(). prefetch_geometry() has to order the side stream behind that work
    (PointHierarchy.readyEvent_), not merely behind the reset: every step must reproduce the inline result although the
    points arrive through asynchronous copies enqueued right before the prefetch."""
    import torch
    from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder
    rng = np.random.default_rng(12)
    host = []
    for n_per, seed in ((60000, 51), (50000, 52), (64000, 53), (40000, 54)):
        pts, bids = make_cloud(n_per, 2, seed, "uniform", True)
        host.append((torch.from_numpy(pts).pin_memory(), torch.from_numpy(bids).pin_memory(),
                     torch.from_numpy(rng.random((len(pts), 1), dtype=np.float32)).pin_memory(),
                     torch.from_numpy(rng.random((len(pts), 16), dtype=np.float32)).cuda()))
    torch.manual_seed(5)
    builder = _builder(protocol, KDEWindow=0.2, relativeRadius=True)
    R = 0.03

    def upload(b):
        hp, hb, hf, og = host[b]
        P, Bi = hp.to("cuda", non_blocking=True), hb.to("cuda", non_blocking=True)
        F = hf.to("cuda", non_blocking=True).requires_grad_(True)
        return PointHierarchy(P, F, Bi, [0.05], "PH", 2, True), F, og   # one Poisson level: more work before the event

    def conv(ph, F, og):
        for p in builder.parameters():
            p.grad = None
        out = builder.create_convolution("Conv", ph, 0, F, 1, R, outNumFeatures=16, multiFeatureConv=True)
        out.backward(og)
        return out

    refs = []
    for b in range(len(host)):                     # inline path
        builder.reset()
        ph, F, og = upload(b)
        out = conv(ph, F, og)
        refs.append((out.detach().clone(), next(iter(builder.cacheNeighs_.values()))[1].clone()))
    torch.cuda.synchronize()
    order = [0, 1, 2, 3, 2, 0, 3, 1]
    builder.reset()
    nxt = upload(order[0])
    builder.prefetch_geometry(nxt[0], 0, R)
    for step, b in enumerate(order):
        ph, F, og = nxt
        builder.reset()                            # installs batch b's geometry
        assert len(builder.cacheNeighs_) == 1
        out = conv(ph, F, og)
        if step + 1 < len(order):
            nxt = upload(order[step + 1])          # AFTER the reset, on the calling stream, asynchronous copies
            builder.prefetch_geometry(nxt[0], 0, R)
        assert torch.equal(next(iter(builder.cacheNeighs_.values()))[1], refs[b][1]), (step, b)
        assert torch.equal(out.detach(), refs[b][0]), (step, b)
    torch.cuda.synchronize()
    if protocol == "ops":   # (the op-by-op protocol orders its side stream behind the hierarchy's own event: recorded from the
        assert nxt[0].readyEvent_ is not None and builder.resetEvent_ is not None   # first such prefetch on, MCConvBuilder._SIDE_EVENTS)


def test_geometry_started_ahead_is_not_served_to_another_hierarchy(mc):
    """prefetch_step() / prefetch_geometry() file what they build under the reference's cache keys (hierarchy NAME, levels,
    radii). A different hierarchy of the same name used after the next reset() must not be served those lists: the builder
    checks a parked geometry against the tensors it was built from before its first use."""
    import torch
    import mccnn_amd.MCConvBuilder as MB
    from mccnn_amd import native
    from tests.helpers import make_cloud
    if not native.side_streams_available():
        pytest.skip("prefetch_step() needs the torch extension (side streams)")
    pa, ba = make_cloud(3000, 3, 11, "clustered", True)
    pb, bb = make_cloud(3000, 3, 12, "uniform", True)
    PA, BA = torch.from_numpy(pa).cuda(), torch.from_numpy(ba).cuda()
    PB, BB = torch.from_numpy(pb).cuda(), torch.from_numpy(bb).cuda()
    radii = [0.1, 0.4]

    def net(builder, ph):
        f = torch.ones((ph.points_[0].shape[0], 1), device="cuda")
        o1 = builder.create_convolution("c1", ph, 0, f, 1, 0.2, ph, 1, True, 16)
        o2 = builder.create_convolution("c2", ph, 1, o1, 16, 0.8, ph, 2, False)
        o3 = builder.create_convolution("c3", ph, 1, o1, 16, 0.4, ph, 1, False)
        o4 = builder.create_convolution("c4", ph, 0, f, 1, 0.1, ph, 0, True, 8)
        o5 = builder.create_convolution("c5", ph, 1, o1, 16, 0.8, ph, 1, False)
        return [o1, o2, o3, o4, o5]

    torch.manual_seed(0)
    b = MB.ConvolutionBuilder(KDEWindow=0.2)
    phA = MB.PointHierarchy(PA, torch.ones((len(pa), 1), device="cuda"), BA, radii, "PH", 3)
    phB = MB.PointHierarchy(PB, torch.ones((len(pb), 1), device="cuda"), BB, radii, "PH", 3)
    for _ in range(2):   # the builder learns the graph
        b.reset()
        refA = [o.detach().clone() for o in net(b, phA)]
    b.reset()
    refB = [o.detach().clone() for o in net(b, phB)]
    # geometry of A started ahead ... and then B (same name, same levels, same radii) is what the next step convolves
    b.reset()
    assert b.prefetch_step(phA) == 5
    net(b, phB)
    b.reset()
    got = net(b, phB)
    for g, r in zip(got, refB):
        assert torch.equal(g, r)
    # ... and A itself is served what was built for it
    b.reset()
    assert b.prefetch_step(phA) == 5
    net(b, phB)
    b.reset()
    used = {k: g for k, g in b.cacheGeo_.items()}
    got = net(b, phA)
    for g, r in zip(got, refA):
        assert torch.equal(g, r)
    assert all(b.cacheGeo_[k] is g for k, g in used.items())   # nothing was rebuilt
    torch.cuda.synchronize()
