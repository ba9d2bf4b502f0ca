"""Generates tests/golden/batcher.npz by running the reference's own loader (utils/DataSet.py `DataSet.get_next_batch`
with its sampling / augmentation helpers, imported from /root/reference in the build container -- the reference cannot
travel, the vectors can) over small seeded in-memory models. For every configuration the fixture stores each batch of one
epoch and the next number the reference's RandomState produces afterwards, so mccnn_amd.batcher.RaggedBatcher is pinned
on outputs AND on generator state.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_batcher.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference/utils")
from DataSet import DataSet  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def make_models(seed, count, lo, hi, nfeat, nlab):
    g = np.random.default_rng(seed)
    models = []
    for _ in range(count):
        n = int(g.integers(lo, hi))
        pts = g.random((n, 3)) * np.array([2.0, 1.0, 1.5])
        nrm = g.normal(size=(n, 3))
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        feats = g.random((n, nfeat)) if nfeat else None
        labels = g.random((n, nlab)) if nlab else None
        models.append(dict(pts=pts, normals=nrm, features=feats, labels=labels))
    return models


class MemDataSet(DataSet):
    def __init__(self, models, categories, **kw):
        DataSet.__init__(self, **kw)
        self.models = models
        self.fileList_ = ["m%d" % i for i in range(len(models))]
        self.numPts_ = [len(m["pts"]) for m in models]
        self.categories_ = categories

    def _load_model_from_disk_(self, modelPath):
        m = self.models[int(modelPath[1:])]
        return m["pts"], m["normals"], m["features"], m["labels"]


CONFIGS = [
    # name, models(seed,count,lo,hi,nfeat,nlab), ctor kwargs, repeatModelInBatch
    ("all_protocols", (1, 10, 60, 160, 2, 1), dict(numPoints=48, ptDropOut=0.9, batchSize=4, allowedSamplings=[0, 1, 2, 3, 4],
                                                    useCategories=True, pointCategories=False, seed=5), False),
    ("max_pts", (2, 9, 40, 200, 0, 0), dict(numPoints=0, ptDropOut=0.8, batchSize=4, allowedSamplings=[0, 1], maxPtsxBatch=330,
                                            useCategories=True, pointCategories=True, seed=7), False),
    ("augment", (3, 6, 50, 90, 6, 3), dict(numPoints=32, ptDropOut=1.0, batchSize=3, allowedSamplings=[0, 2], augment=True,
                                           augmentMainAxis=2, augmentSmallRotations=True, augmentedFeatures=[0, 3],
                                           augmentedLabels=[0], useCategories=False, pointCategories=False, seed=9), False),
    ("select_first_repeat", (4, 5, 80, 120, 1, 0), dict(numPoints=40, ptDropOut=0.95, batchSize=3, allowedSamplings=[0],
                                                         uniformSelectFirst=True, useCategories=True, pointCategories=False,
                                                         seed=11), True),
]

out = {"configs": np.array([c[0] for c in CONFIGS])}
for name, mspec, kw, repeat in CONFIGS:
    models = make_models(*mspec)
    cats = [int(i % 3) for i in range(len(models))]
    ds = MemDataSet(models, cats, pointFeatures=mspec[4] > 0, pointLabels=mspec[5] > 0, pointNormals=True, **kw)
    ds.start_iteration()
    b = 0
    while ds.has_more_batches():
        num, pts, bids, feats, labels, cat, paths = ds.get_next_batch(repeat)
        p = "%s_b%d_" % (name, b)
        out[p + "num"] = np.array(num)
        out[p + "pts"], out[p + "bids"], out[p + "feats"] = np.asarray(pts), np.asarray(bids), np.asarray(feats)
        if labels is not None:
            out[p + "labels"] = np.asarray(labels)
        if cat is not None:
            out[p + "cat"] = np.asarray(cat)
        out[p + "ids"] = np.array([int(s[1:]) for s in paths], dtype=np.int64)
        b += 1
    out[name + "_batches"] = np.array(b)
    out[name + "_next"] = np.array(ds.randomState_.random_sample())
np.savez_compressed(os.path.join(HERE, "batcher.npz"), **out)
print("wrote", {c[0]: int(out[c[0] + "_batches"]) for c in CONFIGS}, os.path.getsize(os.path.join(HERE, "batcher.npz")), "bytes")
