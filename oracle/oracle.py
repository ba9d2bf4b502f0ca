"""ctypes/NumPy front-end of the CPU oracle (oracle/mccnn_oracle.cpp).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never by the product package (mccnn_amd/).
PARITY UNPINNED: see the header of mccnn_oracle.cpp.

Function names and argument orders mirror the reference's Python op surface
(tf_ops/MCConvModuleSrc:20-81) so the parity tests read like calls into the
reference module; arrays are NumPy (float32 / int32) instead of TF tensors.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int)


def build(force=False):
    """Compile the oracle libraries with g++ (no-op when up to date)."""
    src = os.path.join(_DIR, "mccnn_oracle.cpp")
    outs = [os.path.join(_DIR, n) for n in ("liboracle.so", "liboracle_omp.so")]
    if force or any((not os.path.exists(o)) or os.path.getmtime(o) < os.path.getmtime(src) for o in outs):
        subprocess.check_call(["make", "-C", _DIR, "-s", "-B"])


class Oracle:
    """One loaded oracle library. omp=False -> sequential checker; omp=True -> timing build."""

    def __init__(self, omp=False):
        build()
        self.lib = C.CDLL(os.path.join(_DIR, "liboracle_omp.so" if omp else "liboracle.so"))
        self.lib.orc_cell_offsets.restype = _i
        self.lib.orc_cell_offsets_pool.restype = _i

    # -- helpers -----------------------------------------------------------
    @staticmethod
    def _f32(a):
        return np.ascontiguousarray(a, dtype=np.float32)

    @staticmethod
    def _i32(a):
        return np.ascontiguousarray(a, dtype=np.int32)

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(_f if a.dtype == np.float32 else _i)

    def _chk(self, rc, name):
        if rc != 0:
            raise ValueError("oracle %s failed with code %d" % (name, rc))

    def num_threads(self):
        return int(self.lib.orc_num_threads())

    def get_block_size(self):
        return int(self.lib.orc_block_size())

    def cell_offsets(self):
        return np.ctypeslib.as_array(self.lib.orc_cell_offsets(), shape=(27, 3)).copy()

    def cell_offsets_pool(self):
        return np.ctypeslib.as_array(self.lib.orc_cell_offsets_pool(), shape=(27, 3)).copy()

    # -- op surface (MCConvModuleSrc) ---------------------------------------
    def compute_aabb(self, inPts, inBatchIds, batchSize, scaleInv=True):
        pts, bids = self._f32(inPts), self._i32(inBatchIds).reshape(-1)
        mn = np.empty((batchSize, 3), np.float32)
        mx = np.empty((batchSize, 3), np.float32)
        self._chk(self.lib.orc_compute_aabb(self._p(pts), self._p(bids), len(pts), batchSize, int(scaleInv),
                                            self._p(mn), self._p(mx)), "compute_aabb")
        return mn, mx

    def num_cells(self, aabbMin, aabbMax, batchSize, cellSize, scaleInv):
        mn, mx = self._f32(aabbMin), self._f32(aabbMax)
        return int(self.lib.orc_num_cells(self._p(mn), self._p(mx), batchSize, C.c_float(cellSize), int(scaleInv)))

    def sort_points_step1(self, inPts, inBatchIds, aabbMin, aabbMax, batchSize, cellSize, scaleInv):
        pts, bids = self._f32(inPts), self._i32(inBatchIds).reshape(-1)
        mn, mx = self._f32(aabbMin), self._f32(aabbMax)
        nc = self.num_cells(mn, mx, batchSize, cellSize, scaleInv)
        keys = np.empty(len(pts), np.int32)
        idx = np.empty(len(pts), np.int32)
        self._chk(self.lib.orc_sort_step1(self._p(pts), self._p(bids), self._p(mn), self._p(mx), len(pts),
                                          batchSize, nc, self._p(keys), self._p(idx)), "sort_step1")
        return keys, idx

    def sort_points_step2(self, inPts, inBatchIds, inFeatures, keys, indexs, aabbMin, aabbMax, batchSize,
                          cellSize, scaleInv):
        pts, bids = self._f32(inPts), self._i32(inBatchIds).reshape(-1)
        feats = self._f32(inFeatures)
        mn, mx = self._f32(aabbMin), self._f32(aabbMax)
        keys, indexs = self._i32(keys), self._i32(indexs)
        nc = self.num_cells(mn, mx, batchSize, cellSize, scaleInv)
        n, F = feats.shape
        oP = np.empty_like(pts)
        oB = np.empty(n, np.int32)
        oF = np.empty_like(feats)
        cells = np.empty((batchSize, nc, nc, nc, 2), np.int32)
        self._chk(self.lib.orc_sort_step2(self._p(pts), self._p(bids), self._p(feats), self._p(keys),
                                          self._p(indexs), n, F, batchSize, nc, self._p(oP), self._p(oB),
                                          self._p(oF), self._p(cells)), "sort_step2")
        return oP, oB.reshape(-1, 1), oF, cells

    def sort_points_step2_grad(self, indexs, ptsGrad, featGrad):
        return self.sort_features_back(ptsGrad, indexs), self.sort_features_back(featGrad, indexs)

    def sort_features(self, inFeatures, indexs):
        # MCConvModuleSrc:35-36 -> SortFeaturesBackGrad: out[idx[i]] = in[i]
        f, idx = self._f32(inFeatures), self._i32(indexs)
        out = np.empty_like(f)
        self._chk(self.lib.orc_permute_scatter(self._p(f), self._p(idx), f.shape[0], f.shape[1], self._p(out),
                                               f.shape[0], 0), "sort_features")
        return out

    def sort_features_back(self, inFeatures, indexs):
        # MCConvModuleSrc:41-42 -> SortFeaturesBack: out[i] = in[idx[i]]
        f, idx = self._f32(inFeatures), self._i32(indexs)
        out = np.empty_like(f)
        self._chk(self.lib.orc_permute_gather(self._p(f), self._p(idx), f.shape[0], f.shape[1], self._p(out)),
                  "sort_features_back")
        return out

    def transform_indexs(self, inIndexs, inNewPositions):
        a, b = self._i32(inIndexs), self._i32(inNewPositions)
        out = np.empty(len(a), np.int32)
        self._chk(self.lib.orc_transform_indexs(self._p(a), len(a), self._p(b), len(b), self._p(out)),
                  "transform_indexs")
        return out

    def find_neighbors(self, inPts, inBatchIds, inPts2, cellIndexs, aabbMin, aabbMax, radius, batchSize, scaleInv):
        c, cb = self._f32(inPts), self._i32(inBatchIds).reshape(-1)
        p2, cells = self._f32(inPts2), self._i32(cellIndexs)
        mn, mx = self._f32(aabbMin), self._f32(aabbMax)
        nc = cells.shape[1]
        m = len(c)
        start = np.empty(m, np.int32)
        tot = C.c_int(0)
        args = (self._p(c), self._p(cb), m, self._p(p2), self._p(cells), self._p(mn), self._p(mx), batchSize,
                nc, C.c_float(radius), int(scaleInv))
        self._chk(self.lib.orc_find_neighbors_count(*args, self._p(start), C.byref(tot)), "find_neighbors_count")
        packed = np.empty((tot.value, 2), np.int32)
        self._chk(self.lib.orc_find_neighbors_fill(*args, self._p(start), self._p(packed)), "find_neighbors_fill")
        return start.reshape(-1, 1), packed

    def compute_pdf(self, inPts, inBatchIds, aabbMin, aabbMax, startIndexs, neighbors, window, radius, batchSize,
                    scaleInv, exactCount=False):
        """exactCount: see orc_compute_pdf -- the reference's float subtraction of the row offsets breaks beyond 2^24
        edges; False keeps the reference expression."""
        p, b = self._f32(inPts), self._i32(inBatchIds).reshape(-1)
        mn, mx = self._f32(aabbMin), self._f32(aabbMax)
        st, pk = self._i32(startIndexs).reshape(-1), self._i32(neighbors)
        pdfs = np.empty((len(pk), 1), np.float32)
        self._chk(self.lib.orc_compute_pdf(self._p(p), self._p(b), self._p(st), len(st), self._p(pk), len(pk),
                                           self._p(mn), self._p(mx), C.c_float(window), C.c_float(radius),
                                           int(scaleInv), self._p(pdfs), int(bool(exactCount))), "compute_pdf")
        return pdfs

    def poisson_sampling(self, inPts, inBatchIds, cellIndexs, aabbMin, aabbMax, radius, batchSize, scaleInv):
        p, b = self._f32(inPts), self._i32(inBatchIds).reshape(-1)
        cells = self._i32(cellIndexs)
        mn, mx = self._f32(aabbMin), self._f32(aabbMax)
        n, nc = len(p), cells.shape[1]
        oP = np.empty((n, 3), np.float32)
        oB = np.empty(n, np.int32)
        oI = np.empty(n, np.int32)
        s = C.c_int(0)
        self._chk(self.lib.orc_poisson_sampling(self._p(p), self._p(b), n, self._p(cells), self._p(mn),
                                                self._p(mx), batchSize, nc, C.c_float(radius), int(scaleInv),
                                                self._p(oP), self._p(oB), self._p(oI), C.byref(s)),
                  "poisson_sampling")
        s = s.value
        return oP[:s].copy(), oB[:s].reshape(-1, 1).copy(), oI[:s].copy()

    def get_sampled_features(self, inSampledIndexs, pInFeatures):
        return self.sort_features_back_rows(pInFeatures, inSampledIndexs)

    def sort_features_back_rows(self, feats, idx):
        f, idx = self._f32(feats), self._i32(idx)
        out = np.empty((len(idx), f.shape[1]), np.float32)
        self._chk(self.lib.orc_permute_gather(self._p(f), self._p(idx), len(idx), f.shape[1], self._p(out)),
                  "gather")
        return out

    def get_sampled_features_grad(self, inSampledIndexs, pInFeatures, grads):
        idx, g = self._i32(inSampledIndexs), self._f32(grads)
        n, F = np.asarray(pInFeatures).shape
        out = np.empty((n, F), np.float32)
        self._chk(self.lib.orc_permute_scatter(self._p(g), self._p(idx), len(idx), F, self._p(out), n, 1),
                  "get_sampled_features_grad")
        return out

    def _conv_args(self, inPts, inFeatures, inBatchIds, inPDFs, inSamplePts, neighStartIndexs, packedNeighs,
                   aabbMin, aabbMax, weights1, weights2, weightsOut, biases1, biases2, biasesOut):
        a = dict(
            pts=self._f32(inPts), feats=self._f32(inFeatures), bids=self._i32(inBatchIds).reshape(-1),
            pdfs=self._f32(inPDFs).reshape(-1), smp=self._f32(inSamplePts),
            st=self._i32(neighStartIndexs).reshape(-1), pk=self._i32(packedNeighs),
            mn=self._f32(aabbMin), mx=self._f32(aabbMax),
            w1=self._f32(weights1).reshape(-1), b1=self._f32(biases1).reshape(-1),
            w2=self._f32(weights2).reshape(-1), b2=self._f32(biases2).reshape(-1),
            w3=self._f32(weightsOut).reshape(-1), b3=self._f32(biasesOut).reshape(-1))
        return a

    def spatial_conv(self, inPts, inFeatures, inBatchIds, inPDFs, inSamplePts, neighStartIndexs, packedNeighs,
                     aabbMin, aabbMax, weights1, weights2, weightsOut, biases1, biases2, biasesOut,
                     numOutFeatures, combin, batchSize, radius, scaleInv, avg):
        a = self._conv_args(inPts, inFeatures, inBatchIds, inPDFs, inSamplePts, neighStartIndexs, packedNeighs,
                            aabbMin, aabbMax, weights1, weights2, weightsOut, biases1, biases2, biasesOut)
        n, Fin = a["feats"].shape
        m, e = len(a["smp"]), len(a["pk"])
        outF = numOutFeatures if combin else Fin
        out = np.empty((m, outF), np.float32)
        p = self._p
        self._chk(self.lib.orc_spatial_conv_fwd(
            p(a["pts"]), p(a["feats"]), p(a["bids"]), p(a["pdfs"]), p(a["smp"]), p(a["st"]), p(a["pk"]),
            p(a["mn"]), p(a["mx"]), p(a["w1"]), p(a["b1"]), p(a["w2"]), p(a["b2"]), p(a["w3"]), p(a["b3"]),
            n, m, e, Fin, numOutFeatures, int(combin), C.c_float(radius), int(scaleInv), int(avg), p(out)),
            "spatial_conv")
        return out

    def spatial_conv_grad(self, inPts, inFeatures, inBatchIds, inPDFs, inSamplePts, neighStartIndexs,
                          packedNeighs, aabbMin, aabbMax, weights1, weights2, weightsOut, biases1, biases2,
                          biasesOut, outGrad, numOutFeatures, combin, batchSize, radius, scaleInv, avg):
        a = self._conv_args(inPts, inFeatures, inBatchIds, inPDFs, inSamplePts, neighStartIndexs, packedNeighs,
                            aabbMin, aabbMax, weights1, weights2, weightsOut, biases1, biases2, biasesOut)
        og = self._f32(outGrad)
        n, Fin = a["feats"].shape
        m, e = len(a["smp"]), len(a["pk"])
        nn = len(a["b1"])
        fg = np.empty((n, Fin), np.float32)
        dw1, db1 = np.empty(3 * nn, np.float32), np.empty(nn, np.float32)
        dw2, db2 = np.empty(8 * nn, np.float32), np.empty(nn, np.float32)
        dw3, db3 = np.empty(8 * nn, np.float32), np.empty(nn, np.float32)
        p = self._p
        self._chk(self.lib.orc_spatial_conv_bwd(
            p(a["pts"]), p(a["feats"]), p(a["bids"]), p(a["pdfs"]), p(a["smp"]), p(a["st"]), p(a["pk"]),
            p(a["mn"]), p(a["mx"]), p(a["w1"]), p(a["b1"]), p(a["w2"]), p(a["b2"]), p(a["w3"]), p(a["b3"]),
            p(og), n, m, e, Fin, numOutFeatures, int(combin), C.c_float(radius), int(scaleInv), int(avg),
            p(fg), p(dw1), p(db1), p(dw2), p(db2), p(dw3), p(db3)), "spatial_conv_grad")
        # same flat->declared shapes as the reference variables (MCConvBuilder.py:407-419)
        return fg, dw1.reshape(3, nn), db1, dw2.reshape(8, nn), db2, dw3.reshape(8, nn), db3
