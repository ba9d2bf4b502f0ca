"""Ragged batcher of the reference's data loader (utils/DataSet.py:711-842 `get_next_batch`, with
`start_iteration` / `has_more_batches` `:695-708`, `_uniform_sampling_` `:314-361` and `_augment_data_rot_` `:252-311`).

Produces the flattened ragged batch the op chain consumes: points [N,3], batch ids [N,1], features [N,F],
labels, categories. It works on models that are already in memory (file formats and the on-disk cache are out of scope)
and makes the same decisions, in the same order, with the same `numpy.random.RandomState` consumption as the reference
class -- same batches, same generator state afterwards (pinned by tests/golden/batcher.npz, produced by the reference's
own `DataSet`). The non-uniform protocols are the vectorised ones of mccnn_amd.sampling; batch ids and constant features
are built with array operations instead of per-point Python lists. Host-side NumPy only: this module feeds the GPU
path, it is not part of it.
"""
import time

import numpy as np

from . import sampling


class RaggedBatcher:
    """In-memory counterpart of utils/DataSet.py's `DataSet`.

    models: sequence of dicts with keys `pts` [n,3] and optionally `normals` [n,3], `features` [n,m], `labels` [n,l];
    categories: optional per-model ints. The remaining arguments have the reference's names and meaning
    (DataSet.py:81-137); `get_next_batch` returns the reference's 7-tuple with model ids in place of file paths."""

    def __init__(self, models, numPoints, ptDropOut, batchSize, allowedSamplings, categories=None, pointCategories=False,
                 maxPtsxBatch=0, augment=False, augmentMainAxis=1, augmentSmallRotations=False, uniformSelectFirst=False,
                 augmentedFeatures=(), augmentedLabels=(), seed=None):
        if not (0 <= augmentMainAxis < 3):
            raise RuntimeError('Invalid augmentMainAxis')
        if not all(0 <= s < 5 for s in allowedSamplings):
            raise RuntimeError('Invalid sampling protocol')
        self.models_ = list(models)
        self.pointNormals_ = all(m.get("normals") is not None for m in self.models_) and len(self.models_) > 0
        self.pointFeatures_ = all(m.get("features") is not None for m in self.models_) and len(self.models_) > 0
        self.pointLabels_ = all(m.get("labels") is not None for m in self.models_) and len(self.models_) > 0
        if (3 in list(allowedSamplings) or 4 in list(allowedSamplings)) and not self.pointNormals_:
            raise RuntimeError('The dataset should contain normals in order to use the sampling protocols '
                               'lambert and occlusion')
        for lst in (augmentedFeatures, augmentedLabels):
            if any(lst[i + 1] - lst[i] < 3 for i in range(max(len(lst) - 1, 0))):
                raise RuntimeError('The groups of 3 features/labels to augment should not overlap ')
        self.numPoints_ = numPoints
        self.ptDropOut_ = ptDropOut
        self.useCategories_ = categories is not None
        self.pointCategories_ = pointCategories
        self.categories_ = list(categories) if categories is not None else []
        self.batchSize_ = batchSize
        self.allowedSamplings_ = allowedSamplings
        self.maxPtsxBatch_ = maxPtsxBatch
        self.augment_ = augment
        self.augmentMainAxis_ = augmentMainAxis
        self.augmentSmallRotations_ = augmentSmallRotations
        self.augmentedFeatures_ = list(augmentedFeatures)
        self.augmentedLabels_ = list(augmentedLabels)
        self.uniformSelectFirst_ = uniformSelectFirst
        self.numPts_ = [len(m["pts"]) for m in self.models_]
        self.randomSelection_ = []
        self.iterator_ = 0
        self.randomState_ = np.random.RandomState(seed if seed is not None else int(time.time()))

    # ------------------------------------------------------------------ iteration protocol (DataSet.py:649-708)
    def get_num_models(self):
        return len(self.models_)

    def start_iteration(self):
        self.randomSelection_ = self.randomState_.permutation(len(self.models_))
        self.iterator_ = 0

    def has_more_batches(self):
        return self.iterator_ < len(self.randomSelection_)

    # ------------------------------------------------------------------ DataSet.py:252-311
    def _augment_data_rot_(self, inData, mainRotAxis=1, smallRotations=False, inRotationMatrix=None):
        rotationMatrix = inRotationMatrix
        if inRotationMatrix is None:
            rotationAngle = self.randomState_.uniform() * 2.0 * np.pi
            c, s = np.cos(rotationAngle), np.sin(rotationAngle)
            if mainRotAxis == 0:
                rotationMatrix = np.array([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]])
            elif mainRotAxis == 1:
                rotationMatrix = np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])
            else:
                rotationMatrix = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
            if smallRotations:
                a = np.clip(0.06 * self.randomState_.randn(3), -0.18, 0.18)
                Rx = np.array([[1.0, 0.0, 0.0], [0.0, np.cos(a[0]), -np.sin(a[0])], [0.0, np.sin(a[0]), np.cos(a[0])]])
                Ry = np.array([[np.cos(a[1]), 0.0, np.sin(a[1])], [0.0, 1.0, 0.0], [-np.sin(a[1]), 0.0, np.cos(a[1])]])
                Rz = np.array([[np.cos(a[2]), -np.sin(a[2]), 0.0], [np.sin(a[2]), np.cos(a[2]), 0.0], [0.0, 0.0, 1.0]])
                rotationMatrix = np.dot(np.dot(Rz, np.dot(Ry, Rx)), rotationMatrix)
        return np.dot(inData[:, 0:3].reshape((-1, 3)), rotationMatrix), rotationMatrix

    # ------------------------------------------------------------------ DataSet.py:314-361
    def _uniform_sampling_(self, points, inNumPoints, selectFirst, inFeatures=None, inLabels=None, numPoints=0):
        rs = self.randomState_
        if numPoints > 0:
            if selectFirst and inNumPoints >= numPoints:
                choice = rs.choice(int(float(numPoints) * (2.0 - self.ptDropOut_)), numPoints, replace=False)
            else:
                choice = rs.choice(inNumPoints, numPoints, replace=(inNumPoints < numPoints))
        else:
            choice = rs.choice(inNumPoints, int(float(inNumPoints) * self.ptDropOut_), replace=False)
        return (points[choice], None if inFeatures is None else inFeatures[choice],
                None if inLabels is None else inLabels[choice])

    def _sample(self, pts, normals, feats, labels):
        """One model through the protocol the generator picks (DataSet.py:776-796)."""
        rs = self.randomState_
        proto = rs.choice(self.allowedSamplings_)
        if proto == 0:
            return self._uniform_sampling_(pts, len(pts), self.uniformSelectFirst_, feats, labels, self.numPoints_)
        if proto == 1:
            return sampling.sample_split(rs, pts, feats, labels, self.numPoints_)
        if proto == 2:
            return sampling.sample_gradient(rs, pts, feats, labels, self.numPoints_)
        view = sampling.random_view(rs)
        if proto == 3:
            return sampling.sample_lambert(rs, view, pts, normals, feats, labels, self.numPoints_)
        return sampling.sample_occlusion(view, pts, normals, feats, labels, self.numPoints_)

    # ------------------------------------------------------------------ DataSet.py:711-842
    def get_next_batch(self, repeatModelInBatch=False):
        """-> (numModelInBatch, accumPts [N,3], accumBatchIds [N,1], accumFeatures [N,F], accumLabels or None,
        accumCat or None, accumIds). A model that would push the batch beyond maxPtsxBatch leaves its slot -- and,
        because the iterator does not advance, every later slot of this batch -- empty: batch ids are SLOT indices, as
        in the reference."""
        pts_l, bid_l, feat_l, lab_l, cat_l, ids = [], [], [], [], [], []
        numModelInBatch = 0
        numPtsInBatch = 0
        for i in range(self.batchSize_):
            if self.iterator_ >= len(self.randomSelection_):
                continue
            idx = self.randomSelection_[self.iterator_]
            model = self.models_[idx]
            n_model = self.numPts_[idx]
            if not (self.maxPtsxBatch_ == 0 or (numPtsInBatch + n_model) <= self.maxPtsxBatch_):
                continue
            pts, feats, labels = self._sample(model["pts"], model.get("normals"), model.get("features"),
                                              model.get("labels"))
            if self.augment_:
                pts, rot = self._augment_data_rot_(pts, self.augmentMainAxis_, self.augmentSmallRotations_)
                if self.pointFeatures_:
                    for blk in self.augmentedFeatures_:
                        feats[:, blk:blk + 3], _ = self._augment_data_rot_(feats[:, blk:blk + 3], self.augmentMainAxis_,
                                                                           self.augmentSmallRotations_, rot)
                if self.pointLabels_:
                    for blk in self.augmentedLabels_:
                        labels[:, blk:blk + 3], _ = self._augment_data_rot_(labels[:, blk:blk + 3], self.augmentMainAxis_,
                                                                            self.augmentSmallRotations_, rot)
            k = len(pts)
            pts_l.append(pts)
            bid_l.append(np.full((k, 1), i, dtype=np.int64))
            feat_l.append(feats if self.pointFeatures_ else np.ones((k, 1), dtype=np.float64))
            if self.pointLabels_:
                lab_l.append(labels)
            if self.useCategories_:
                c = self.categories_[idx]
                cat_l.append(np.full((k, 1), c) if self.pointCategories_ else np.array([c]))
            ids.append(idx)
            numPtsInBatch += n_model
            numModelInBatch += 1
            if not repeatModelInBatch:
                self.iterator_ += 1
        if repeatModelInBatch:
            self.iterator_ += 1
        cat = (lambda l: np.concatenate(l, axis=0) if l else np.array([]))
        accumLabels = cat(lab_l) if self.pointLabels_ else None
        accumCat = cat(cat_l) if self.useCategories_ else None
        return numModelInBatch, cat(pts_l), cat(bid_l), cat(feat_l), accumLabels, accumCat, ids

    def to_device(self, batch, device="cuda"):
        """The tensors the op chain takes (points f32 [N,3], batch ids i32 [N,1], features f32 [N,F]) from one
        `get_next_batch` result."""
        import torch
        _, pts, bids, feats = batch[:4]
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(device)
        return t(pts, np.float32), t(bids, np.int32), t(feats, np.float32)
