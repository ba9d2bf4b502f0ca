"""Vectorised sampling protocols (mccnn_amd/sampling.py) against vectors produced by the reference's own loops
(tests/golden/make_golden_sampling.py): selected points, features, labels and the generator state afterwards."""
import os

import numpy as np
import pytest

from mccnn_amd import sampling as S

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampling.npz"))


@pytest.mark.parametrize("case", range(int(G["num_cases"])))
def test_protocol_matches_reference(case):
    k = "c%03d" % case
    proto, inp, seed, num = G[k + "_meta"]
    seed, num = int(seed), int(num)
    pts, nrm, feats, labels, view = (G[str(inp) + s] for s in ("_pts", "_nrm", "_feats", "_labels", "_view"))
    rs = np.random.RandomState(seed)
    if proto == "split":
        got = S.sample_split(rs, pts, feats, labels, num)
    elif proto == "gradient":
        got = S.sample_gradient(rs, pts, feats, labels, num)
    elif proto == "lambert":
        got = S.sample_lambert(rs, view, pts, nrm, feats, labels, num)
    else:
        got = S.sample_occlusion(view, pts, nrm, feats, labels, num)
    for a, b in zip(got, (G[k + "_oP"], G[k + "_oF"], G[k + "_oL"])):
        assert a.dtype == b.dtype and np.array_equal(a, b)
    assert rs.random_sample() == float(G[k + "_next"])


def test_view_direction_and_room_scale():
    v = S.random_view(np.random.RandomState(5))
    assert abs(np.linalg.norm(v) - 1.0) < 1e-12
    # 100k points: the per-point loops of the reference need ~1 s per protocol, this must stay interactive
    g = np.random.default_rng(0)
    pts = g.random((100000, 3)).astype(np.float32) * np.array([6, 4, 2.8], np.float32)
    import time
    t0 = time.perf_counter()
    p, _, _ = S.sample_gradient(np.random.RandomState(1), pts, None, None, 60000)
    assert p.shape == (60000, 3) and time.perf_counter() - t0 < 1.0
