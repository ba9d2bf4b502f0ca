"""The C-ABI is usable without Python or torch: build examples/capi_example.cpp against libmccnn_hip.so with hipcc, run it
on the GPU and compare its result with the oracle on the identical (LCG-generated) input."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lcg_inputs(n, fin=3, nb=3):
    st = np.uint32(12345)
    vals = []

    def rnd(count):
        nonlocal st
        out = np.empty(count, np.float32)
        s = int(st)
        for k in range(count):
            s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
            out[k] = np.float32(s >> 8) * np.float32(1.0 / 16777216.0)
        st = np.uint32(s)
        return out
    pts = rnd(n * 3).reshape(n, 3)
    feats = (2 * rnd(n * fin) - 1).astype(np.float32).reshape(n, fin)
    w1, w2, w3 = [(rnd(c) - np.float32(0.5)).astype(np.float32) for c in (24 * nb, 64 * nb, 64 * nb)]
    b1, b2, b3 = [(np.float32(0.1) * (rnd(8 * nb) - np.float32(0.5))).astype(np.float32) for _ in range(3)]
    og = (2 * rnd(n * 8) - 1).astype(np.float32).reshape(n, 8)   # the out-gradient follows in the same stream
    return pts, feats, w1, b1, w2, b2, w3, b3, og


def test_c_program_matches_oracle(mc, oracle, tmp_path):
    from mccnn_amd import build
    lib_dir = os.path.dirname(build.LIB)
    exe = str(tmp_path / "capi_example")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "capi_example.cpp"), "-L" + lib_dir, "-lmccnn_hip",
                           "-Wl,-rpath," + lib_dir, "-o", exe])
    n, radius = 2048, 0.1
    dump = str(tmp_path / "dump.bin")
    out = subprocess.check_output([exe, str(n), str(radius), dump], text=True)
    m = re.search(r"nc=(\d+) E=(\d+) out_sum=(\S+) out_abs_sum=(\S+)", out)
    assert m, out
    nc, E, s, a = int(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4))
    pts, feats, w1, b1, w2, b2, w3, b3, og = _lcg_inputs(n)
    bids = np.zeros((n, 1), np.int32)
    mn, mx = oracle.compute_aabb(pts, bids, 1, True)
    k, i = oracle.sort_points_step1(pts, bids, mn, mx, 1, radius, True)
    sp, sb, sf, cells = oracle.sort_points_step2(pts, bids, feats, k, i, mn, mx, 1, radius, True)
    st, pk = oracle.find_neighbors(pts, bids, sp, cells, mn, mx, radius, 1, True)
    pdf = oracle.compute_pdf(sp, sb, mn, mx, st, pk, 0.2, radius, 1, True)
    ref = oracle.spatial_conv(sp, sf, sb, pdf, pts, st, pk, mn, mx, w1, w2, w3, b1, b2, b3, 8, True, 1, radius, True, True)
    assert nc == cells.shape[1] and E == len(pk)
    assert abs(a - np.abs(ref.astype(np.float64)).sum()) <= 1e-4 * np.abs(ref).sum()
    assert abs(s - ref.astype(np.float64).sum()) <= 1e-4 * np.abs(ref).sum()
    # forward AND backward of the stand-alone C path, every tensor element by element against the oracle
    rg = oracle.spatial_conv_grad(sp, sf, sb, pdf, pts, st, pk, mn, mx, w1, w2, w3, b1, b2, b3, og, 8, True, 1, radius, True, True)
    fg_ref = oracle.sort_points_step2_grad(i, np.zeros_like(sp), rg[0])[1]
    raw = np.fromfile(dump, np.float32)
    refs = [("out", ref), ("feat_grad", fg_ref), ("dw1", rg[1]), ("db1", rg[2]), ("dw2", rg[3]), ("db2", rg[4]), ("dw3", rg[5]),
            ("db3", rg[6])]
    assert raw.size == sum(np.asarray(r).size for _, r in refs)
    from tests.helpers import assert_float_close
    o = 0
    for name, r in refs:
        r = np.asarray(r, np.float32)
        assert_float_close(raw[o:o + r.size].reshape(r.shape), r, 1e-4, name)
        o += r.size
    # ... and the native step executor (mccnn_geometry_* / mccnn_conv_*) from plain C agrees with the op-level calls
    m2 = re.search(r"executor_vs_ops_max_rel=(\S+)", out)
    assert m2 and float(m2.group(1)) <= 2e-5, out
