"""Generates tests/golden/sampling.npz by running the reference's own sampling loops (utils/DataSet.py:364-646, imported
from /root/reference in the build container -- the reference cannot travel, the vectors can). Inputs are seeded random
clouds; for every case the fixture stores the selected points / features / labels and the next number the reference's
RandomState produces afterwards, so the vectorised port is pinned on outputs AND on generator state.

    python tests/golden/make_golden_sampling.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference/utils")
from DataSet import DataSet  # noqa: E402


class _DS(DataSet):
    def _load_model_from_disk_(self, modelPath):
        raise NotImplementedError


def _ref(seed):
    d = _DS.__new__(_DS)
    d.randomState_ = np.random.RandomState(seed)
    return d


out = {}
case = 0
for dt in (np.float32, np.float64):
    for seed in (3, 11):
        g = np.random.default_rng(seed)
        n = int(g.integers(200, 600))
        pts = (g.random((n, 3)) * np.array([3.0, 1.0, 2.0])).astype(dt)
        nrm = g.normal(size=(n, 3))
        nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(dt)
        feats = g.random((n, 2)).astype(dt)
        labels = g.integers(0, 5, (n, 1))
        vr = np.random.RandomState(seed + 100)
        view = (vr.rand(3) * 2.0) - 1.0
        view = view / np.linalg.norm(view)
        for num in (0, n // 3, 2 * n + 7):
            for proto in ("split", "gradient", "lambert", "occlusion"):
                d = _ref(seed)
                if proto == "split":
                    r = d._non_uniform_sampling_split_(pts, n, feats, labels, num)
                elif proto == "gradient":
                    r = d._non_uniform_sampling_gradient_(pts, n, feats, labels, num)
                elif proto == "lambert":
                    r = d._non_uniform_sampling_lambert_(view, pts, nrm, n, feats, labels, num)
                else:
                    r = d._non_uniform_sampling_occlusion_(view, pts, nrm, n, feats, labels, num)
                k = "c%03d" % case
                inp = "in_%s_%d" % (dt.__name__, seed)  # inputs are stored once per (dtype, seed)
                out[k + "_meta"] = np.array([proto, inp, str(seed), str(num)])
                out[inp + "_pts"], out[inp + "_nrm"], out[inp + "_feats"], out[inp + "_labels"], out[inp + "_view"] = pts, nrm, feats, labels, view
                out[k + "_oP"], out[k + "_oF"], out[k + "_oL"] = np.asarray(r[0]), np.asarray(r[1]), np.asarray(r[2])
                out[k + "_next"] = np.array(d.randomState_.random_sample())
                case += 1
out["num_cases"] = np.array(case)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sampling.npz"), **out)
print("wrote", case, "cases")
