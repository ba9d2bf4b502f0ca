"""TEST INFRASTRUCTURE: the CPU oracle behind the reference's op names, on CPU torch tensors, with the reference's
gradient wiring (tf_ops/MCConvModuleSrc:30-81) as torch.autograd.Functions. Handing this object to
mccnn_amd.MCConvBuilder's classes (`ops=`) runs the identical network graph through the checker, so whole models can be
compared with the HIP path end to end. Never imported by the product package."""
import numpy as np
import torch


def _n(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else t


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


class OracleOps:
    def __init__(self, oracle):
        self.o = oracle
        ops = self

        class SortStep2(torch.autograd.Function):
            @staticmethod
            def forward(ctx, pts, bids, feats, keys, idx, mn, mx, B, cs, si):
                r = ops.o.sort_points_step2(_n(pts), _n(bids), _n(feats), _n(keys), _n(idx), _n(mn), _n(mx), B, cs, si)
                ctx.idx, ctx.fshape = _n(idx), tuple(feats.shape)
                oP, oB, oF, cells = (_t(x) for x in r)
                ctx.mark_non_differentiable(oB, cells)
                return oP, oB, oF, cells

            @staticmethod
            def backward(ctx, gP, gB, gF, gC):
                n = len(ctx.idx)
                gp = _n(gP) if gP is not None else np.zeros((n, 3), np.float32)
                gf = _n(gF) if gF is not None else np.zeros(ctx.fshape, np.float32)
                rp, rf = ops.o.sort_points_step2_grad(ctx.idx, gp, gf)
                return _t(rp), None, _t(rf), None, None, None, None, None, None, None

        class SortFeatures(torch.autograd.Function):
            @staticmethod
            def forward(ctx, f, idx):
                ctx.idx = _n(idx)
                return _t(ops.o.sort_features(_n(f), ctx.idx))

            @staticmethod
            def backward(ctx, g):
                return _t(ops.o.sort_features_back(_n(g), ctx.idx)), None

        class SortFeaturesBack(torch.autograd.Function):
            @staticmethod
            def forward(ctx, f, idx):
                ctx.idx = _n(idx)
                return _t(ops.o.sort_features_back(_n(f), ctx.idx))

            @staticmethod
            def backward(ctx, g):
                return _t(ops.o.sort_features(_n(g), ctx.idx)), None

        class Sampled(torch.autograd.Function):
            @staticmethod
            def forward(ctx, idx, f):
                ctx.idx, ctx.f = _n(idx), _n(f)
                return _t(ops.o.get_sampled_features(ctx.idx, ctx.f))

            @staticmethod
            def backward(ctx, g):
                return None, _t(ops.o.get_sampled_features_grad(ctx.idx, ctx.f, _n(g)))

        class Conv(torch.autograd.Function):
            @staticmethod
            def forward(ctx, pts, feats, bids, pdfs, smp, st, pk, mn, mx, w1, w2, w3, b1, b2, b3, fout, combin, B, radius,
                        si, avg):
                a = tuple(_n(x) for x in (pts, feats, bids, pdfs, smp, st, pk, mn, mx, w1, w2, w3, b1, b2, b3))
                ctx.a, ctx.attrs, ctx.shapes = a, (fout, combin, B, radius, si, avg), (w1.shape, w2.shape, w3.shape)
                return _t(ops.o.spatial_conv(*a, fout, combin, B, radius, si, avg))

            @staticmethod
            def backward(ctx, g):
                fg, dw1, db1, dw2, db2, dw3, db3 = ops.o.spatial_conv_grad(*ctx.a, _n(g), *ctx.attrs)
                s1, s2, s3 = ctx.shapes
                return (None, _t(fg), None, None, None, None, None, None, None, _t(dw1).reshape(s1), _t(dw2).reshape(s2),
                        _t(dw3).reshape(s3), _t(db1), _t(db2), _t(db3), None, None, None, None, None, None)

        self._step2, self._sf, self._sfb, self._sampled, self._conv = SortStep2, SortFeatures, SortFeaturesBack, Sampled, Conv

    def get_block_size(self):
        return self.o.get_block_size()

    def compute_aabb(self, pts, bids, B, scaleInv=True):
        return tuple(_t(x) for x in self.o.compute_aabb(_n(pts), _n(bids), B, scaleInv))

    def sort_points_step1(self, pts, bids, mn, mx, B, cs, si):
        return tuple(_t(x) for x in self.o.sort_points_step1(_n(pts), _n(bids), _n(mn), _n(mx), B, cs, si))

    def sort_points_step2(self, pts, bids, feats, keys, idx, mn, mx, B, cs, si):
        return self._step2.apply(pts, bids, feats, keys, idx, mn, mx, B, cs, si)

    def sort_features(self, f, idx):
        return self._sf.apply(f, idx)

    def sort_features_back(self, f, idx):
        return self._sfb.apply(f, idx)

    def transform_indexs(self, a, b):
        return _t(self.o.transform_indexs(_n(a), _n(b)))

    def find_neighbors(self, c, cb, p2, cells, mn, mx, radius, B, si):
        return tuple(_t(x) for x in self.o.find_neighbors(_n(c), _n(cb), _n(p2), _n(cells), _n(mn), _n(mx), radius, B, si))

    def compute_pdf(self, p, b, mn, mx, st, pk, window, radius, B, si):
        return _t(self.o.compute_pdf(_n(p), _n(b), _n(mn), _n(mx), _n(st), _n(pk), window, radius, B, si))

    def poisson_sampling(self, p, b, cells, mn, mx, radius, B, si):
        return tuple(_t(x) for x in self.o.poisson_sampling(_n(p), _n(b), _n(cells), _n(mn), _n(mx), radius, B, si))

    def get_sampled_features(self, idx, f):
        return self._sampled.apply(idx, f)

    def spatial_conv(self, pts, feats, bids, pdfs, smp, st, pk, mn, mx, w1, w2, w3, b1, b2, b3, fout, combin, B, radius, si,
                     avg):
        return self._conv.apply(pts, feats, bids, pdfs, smp, st, pk, mn, mx, w1, w2, w3, b1, b2, b3, fout, combin, B,
                                radius, si, avg)
