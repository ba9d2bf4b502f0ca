"""mccnn_amd.batcher.RaggedBatcher against batches produced by the reference's own loader (utils/DataSet.py:711-842,
fixture tests/golden/batcher.npz written by tests/golden/make_golden_batcher.py): every batch of an epoch bit for bit --
points, batch ids (slot indices), features, labels, categories, model order -- and the generator state afterwards."""
import os

import numpy as np
import pytest

from mccnn_amd.batcher import RaggedBatcher

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "batcher.npz")


def make_models(seed, count, lo, hi, nfeat, nlab):  # the generator's models (tests/golden/make_golden_batcher.py)
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


CONFIGS = {
    "all_protocols": ((1, 10, 60, 160, 2, 1), dict(numPoints=48, ptDropOut=0.9, batchSize=4, allowedSamplings=[0, 1, 2, 3, 4],
                                                   pointCategories=False, seed=5), True, False),
    "max_pts": ((2, 9, 40, 200, 0, 0), dict(numPoints=0, ptDropOut=0.8, batchSize=4, allowedSamplings=[0, 1], maxPtsxBatch=330,
                                            pointCategories=True, seed=7), True, False),
    "augment": ((3, 6, 50, 90, 6, 3), dict(numPoints=32, ptDropOut=1.0, batchSize=3, allowedSamplings=[0, 2], augment=True,
                                          augmentMainAxis=2, augmentSmallRotations=True, augmentedFeatures=[0, 3],
                                          augmentedLabels=[0], seed=9), False, False),
    "select_first_repeat": ((4, 5, 80, 120, 1, 0), dict(numPoints=40, ptDropOut=0.95, batchSize=3, allowedSamplings=[0],
                                                        uniformSelectFirst=True, pointCategories=False, seed=11), True, True),
}


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_batches_match_the_reference_loader(name):
    gold = np.load(GOLD)
    mspec, kw, use_cat, repeat = CONFIGS[name]
    models = make_models(*mspec)
    cats = [int(i % 3) for i in range(len(models))] if use_cat else None
    rb = RaggedBatcher(models, categories=cats, **kw)
    rb.start_iteration()
    b = 0
    while rb.has_more_batches():
        num, pts, bids, feats, labels, cat, ids = rb.get_next_batch(repeat)
        p = "%s_b%d_" % (name, b)
        assert num == int(gold[p + "num"])
        assert np.array_equal(np.asarray(ids), gold[p + "ids"])
        for key, val in (("pts", pts), ("bids", bids), ("feats", feats), ("labels", labels), ("cat", cat)):
            if val is None:
                assert (p + key) not in gold.files
                continue
            ref = gold[p + key]
            assert np.asarray(val).shape == ref.shape, (key, np.asarray(val).shape, ref.shape)
            assert np.array_equal(np.asarray(val), ref), (name, b, key)
        b += 1
    assert b == int(gold[name + "_batches"])
    assert rb.randomState_.random_sample() == float(gold[name + "_next"])


def test_point_budget_leaves_later_slots_empty():
    """maxPtsxBatch: the model that does not fit is NOT skipped -- the iterator stays on it, so the rest of the batch stays
    empty and it opens the next batch (DataSet.py:760-762, 828-835); batch ids are slot indices."""
    models = [dict(pts=np.zeros((n, 3))) for n in (100, 100, 250, 100)]
    rb = RaggedBatcher(models, numPoints=0, ptDropOut=1.0, batchSize=4, allowedSamplings=[0], maxPtsxBatch=300, seed=0)
    rb.start_iteration()
    seen = []
    while rb.has_more_batches():
        num, pts, bids, *_ , ids = rb.get_next_batch()
        assert sum(len(models[i]["pts"]) for i in ids) <= 300 and num == len(ids) >= 1
        assert len(np.unique(bids)) == num
        seen += list(ids)
    assert sorted(seen) == [0, 1, 2, 3]
