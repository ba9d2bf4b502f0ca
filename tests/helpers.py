"""Shared input generators and the op-chain driver used by both the oracle and the GPU tests."""
import numpy as np

from mccnn_amd.workloads import make_room, conv_nb  # noqa: F401  (the generators live with the product: bench.py uses them too)


def make_cloud(n_per, B, seed, kind="uniform", ragged=False):
    """Flattened ragged batch: points [N,3] f32, batch ids [N,1] i32, clouds concatenated."""
    rng = np.random.default_rng(seed)
    pts, bids = [], []
    for b in range(B):
        n = n_per if not ragged else max(1, int(n_per * (0.3 + 0.7 * rng.random())))
        if kind == "uniform":
            p = rng.random((n, 3), dtype=np.float32)
        elif kind == "sphere":
            v = rng.normal(size=(n, 3))
            p = (v / np.linalg.norm(v, axis=1, keepdims=True) * (0.5 + 0.1 * b)).astype(np.float32)
        elif kind == "clustered":  # strongly non-uniform: dense blobs + sparse background
            k = n // 2
            c = rng.random((4, 3))
            blob = c[rng.integers(0, 4, k)] + 0.03 * rng.normal(size=(k, 3))
            p = np.concatenate([blob, rng.random((n - k, 3))]).astype(np.float32)
            rng.shuffle(p)
        else:
            raise ValueError(kind)
        pts.append(p + np.float32(0.25 * b))
        bids.append(np.full((len(p), 1), b, np.int32))
    return np.concatenate(pts).astype(np.float32), np.concatenate(bids).astype(np.int32)


def make_mlp(nb, seed, scale=0.5, bias=0.1):
    """Kernel-MLP tensors in the reference's declared shapes (MCConvBuilder.py:407-419)."""
    rng = np.random.default_rng(seed)
    nn = 8 * nb
    u = lambda *s: (scale * (2 * rng.random(s) - 1)).astype(np.float32)
    return dict(w1=u(3, nn), b1=(bias * u(nn)), w2=u(8, nn), b2=(bias * u(nn)), w3=u(8, nn), b3=(bias * u(nn)))


def run_chain(ops, wrap, unwrap, pts, bids, feats, B, radius, scaleInv, window=0.2, fout=8, combin=True, avg=True,
              seed=7, poisson_radius=None, centres=None, centre_bids=None, pdf_kwargs=None):
    """compute_aabb -> sort_step1/2 -> find_neighbors -> compute_pdf -> spatial_conv (+grad) [-> poisson ...],
    i.e. ConvolutionBuilder.create_convolution's op sequence (MCConvBuilder.py:349-427).
    `ops` is a module/object exposing the reference's op names; wrap/unwrap convert numpy<->backend tensors."""
    r = {}
    P, Bi, F = wrap(pts), wrap(bids), wrap(feats)
    mn, mx = ops.compute_aabb(P, Bi, B, scaleInv)
    keys, idx = ops.sort_points_step1(P, Bi, mn, mx, B, radius, scaleInv)
    sP, sB, sF, cells = ops.sort_points_step2(P, Bi, F, keys, idx, mn, mx, B, radius, scaleInv)
    C = P if centres is None else wrap(centres)
    Cb = Bi if centres is None else wrap(centre_bids)
    start, packed = ops.find_neighbors(C, Cb, sP, cells, mn, mx, radius, B, scaleInv)
    pdfs = ops.compute_pdf(sP, sB, mn, mx, start, packed, window, radius, B, scaleInv, **(pdf_kwargs or {}))
    r.update(aabbMin=unwrap(mn), aabbMax=unwrap(mx), keys=unwrap(keys), indexs=unwrap(idx), sortPts=unwrap(sP),
             sortBatchs=unwrap(sB), sortFeatures=unwrap(sF), cellIndexs=unwrap(cells), startIndexs=unwrap(start),
             packedNeighs=unwrap(packed), pdfs=unwrap(pdfs))
    fin = feats.shape[1]
    w = make_mlp(conv_nb(fin, fout, combin), seed)
    r["mlp"] = w
    r["_handles"] = dict(sP=sP, sB=sB, sF=sF, cells=cells, mn=mn, mx=mx, start=start, packed=packed, pdfs=pdfs, C=C,
                         idx=idx)
    if poisson_radius is not None:
        # a second grid at the Poisson radius, like PointHierarchy.__init__ (MCConvBuilder.py:101-116)
        k2, i2 = ops.sort_points_step1(P, Bi, mn, mx, B, poisson_radius, scaleInv)
        p2, b2, f2, c2 = ops.sort_points_step2(P, Bi, F, k2, i2, mn, mx, B, poisson_radius, scaleInv)
        sp, sb, si = ops.poisson_sampling(p2, b2, c2, mn, mx, poisson_radius, B, scaleInv)
        sf = ops.get_sampled_features(si, f2)
        ti = ops.transform_indexs(si, i2)
        r.update(samplePts=unwrap(sp), sampleBatchs=unwrap(sb), sampleIndexs=unwrap(si), sampleFeatures=unwrap(sf),
                 transformedIndexs=unwrap(ti), poissonSortPts=unwrap(p2), poissonCells=unwrap(c2))
    return r


# ---------------------------------------------------------------------------------------------- tolerances
#: Per ELEMENT: |got - ref| <= rtol |ref| + FLOOR max|ref|. The absolute floor is for elements that cancel to ~0: a sum of k
#: terms of size ~max|ref| carries rounding noise of ~1e-7 k^(1/2) max|ref| whatever its own value (measured worst case over
#: every test: 2.4e-6 max|ref|, float atomics in the feature gradient); 1e-5 leaves a factor of four.
ELEMENT_FLOOR = 1e-5


def elementwise_excess(got, ref, rtol=1e-4, floor=ELEMENT_FLOOR):
    """max over the elements of |got - ref| / (rtol |ref| + floor max|ref|); <= 1 passes. Next to every norm-wise bound
    (max|diff| / max|ref| <= rtol) the float tests hold this one, so that "within 1e-4 relative" is true of every value
    that is not noise-level small, not only of the largest ones."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    if ref.size == 0:
        return 0.0
    scale = max(float(np.abs(ref).max()), 1e-30)
    return float((np.abs(got - ref) / (rtol * np.abs(ref) + floor * scale)).max())


def assert_float_close(got, ref, rtol=1e-4, what="", floor=ELEMENT_FLOOR):
    """norm-wise AND element-wise (see elementwise_excess)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    if ref.size == 0:
        return 0.0
    scale = max(float(np.abs(ref).max()), 1e-30)
    err = float(np.abs(got - ref).max() / scale)
    assert err <= rtol, "%s: max |diff| / max |ref| = %.3e > %.1e" % (what, err, rtol)
    ex = elementwise_excess(got, ref, rtol, floor)
    assert ex <= 1.0, "%s: an element is %.2f x outside |d| <= %.0e |ref| + %.0e max|ref|" % (what, ex, rtol, floor)
    return err
