"""HIP path against the committed golden fixtures (tests/golden/chain_*.npz; see make_golden.py for provenance)."""
import glob
import os

import numpy as np
import pytest

from tests.helpers import run_chain, assert_float_close

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-4  # north_star: fp32 features within 1e-4 relative


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "chain_*.npz"))), ids=os.path.basename)
def test_hip_matches_golden(mc, path):
    import torch
    g = np.load(path)
    B, radius, scaleInv, fin, fout, combin, prad = g["attrs"]
    B, scaleInv, fin, fout, combin = int(B), bool(scaleInv), int(fin), int(fout), bool(combin)
    wrap = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    unwrap = lambda t: t.detach().cpu().numpy()
    o = run_chain(mc, wrap, unwrap, g["in_points"], g["in_batch_ids"], g["in_features"], B, float(radius), scaleInv,
                  fout=fout, combin=combin, poisson_radius=float(prad))
    for k in ("keys", "indexs", "cellIndexs", "startIndexs", "packedNeighs", "sampleIndexs", "transformedIndexs",
              "sampleBatchs", "sortBatchs"):
        assert np.array_equal(o[k], g[k]), k
    def rel(a, b):  # norm-wise for the caller, per element here
        return assert_float_close(a, b, RTOL, "golden")
    assert rel(o["pdfs"], g["pdfs"]) <= RTOL
    h = o["_handles"]
    tw = {k: wrap(g["mlp_" + k]).requires_grad_(True) for k in ("w1", "b1", "w2", "b2", "w3", "b3")}
    sF = h["sF"].detach().clone().requires_grad_(True)
    out = mc.spatial_conv(h["sP"], sF, h["sB"], wrap(g["pdfs"]), h["C"], h["start"], h["packed"], h["mn"], h["mx"],
                          tw["w1"], tw["w2"], tw["w3"], tw["b1"], tw["b2"], tw["b3"], fout, combin, B, float(radius),
                          scaleInv, True)
    assert rel(unwrap(out), g["conv_out"]) <= RTOL
    out.backward(wrap(g["out_grad"]))
    for nm, t in (("feat_grad", sF), ("dw1", tw["w1"]), ("db1", tw["b1"]), ("dw2", tw["w2"]), ("db2", tw["b2"]),
                  ("dw3", tw["w3"]), ("db3", tw["b3"])):
        assert rel(unwrap(t.grad), g[nm]) <= RTOL, nm
