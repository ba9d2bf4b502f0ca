#!/usr/bin/env python3
"""Generates the fixtures under tests/golden/. Run in the BUILD container only (it imports the reference from
/root/reference, which does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Fixtures are DATA (inputs and expected outputs), never reference source:
  nonuniform_cloud.npz   a 4096-point non-uniform cloud produced by the REFERENCE's own gradient sampling
                         protocol (utils/DataSet.py:431-492, imported and executed here) from a seeded uniform
                         cloud -- an input fixture for the parity tests;
  builder_trace.json     the op-call trace and variable shapes the REFERENCE's MCConvBuilder / MCClassS graph
                         builder emits (utils/MCConvBuilder.py, models/MCClassS.py executed with recording stubs
                         for tensorflow / MCConvModule / MCNetworkUtils) -- pins the builder counterpart;
  chain_*.npz            outputs of the CPU oracle (oracle/mccnn_oracle.cpp) on fixed inputs. The reference's ops
                         are GPU-only TF1 custom ops and cannot run here, so these are ORACLE outputs
                         (regression pins), not reference outputs: parity stays "unpinned" (see oracle header).
"""
import json
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def nonuniform_cloud():
    sys.path.insert(0, os.path.join(REF, "utils"))
    import DataSet as refds  # the reference module itself
    # no __init__ / loader: only the reference's sampling method and its RNG are needed
    sub = type("FixtureDataSet", (refds.DataSet,), {"_load_model_from_disk_": lambda self, p: None})
    obj = sub.__new__(sub)
    obj.randomState_ = np.random.RandomState(20180601)
    rng = np.random.default_rng(5)
    src = rng.random((20000, 3)).astype(np.float32) * np.array([2.0, 1.0, 0.5], np.float32)
    pts, _, _ = obj._non_uniform_sampling_gradient_(src, len(src), numPoints=4096)
    pts = np.asarray(pts, np.float32)
    np.savez_compressed(os.path.join(HERE, "nonuniform_cloud.npz"), points=pts)
    print("nonuniform_cloud", pts.shape)


class _Sym:
    """Symbolic tensor stand-in for the recording stubs."""

    def __init__(self, name, shape=None):
        self.name, self.shape = name, shape

    def __repr__(self):
        return self.name


def builder_trace():
    trace = []
    tf = types.ModuleType("tensorflow")
    tf.float32 = "float32"
    variables = {}

    def get_variable(name, shape=None, initializer=None):
        variables[name] = list(shape)
        trace.append(["get_variable", name, list(shape)])
        return _Sym(name, list(shape))

    tf.get_variable = get_variable
    tf.add_to_collection = lambda coll, v: trace.append(["add_to_collection", coll, v.name])
    tf.reshape = lambda t, shape: _Sym(t.name + ":reshape", list(shape))
    tf.zeros_initializer = lambda: "zeros"
    tf.shape = lambda t: [None, None]
    tf.ones = lambda shape, dtype=None: _Sym("ones")
    tf.concat = lambda ts, axis: _Sym("concat")
    contrib = types.SimpleNamespace(layers=types.SimpleNamespace(
        variance_scaling_initializer=lambda factor=1.0, mode="FAN_AVG", uniform=True: "vs(%s,%s,%s)" % (factor, mode, uniform)))
    tf.contrib = contrib
    sys.modules["tensorflow"] = tf

    mod = types.ModuleType("MCConvModule")
    counter = [0]

    def rec(name, nout, static_from=None):
        def f(*args):
            counter[0] += 1
            statics = [a for a in args if isinstance(a, (int, float, bool)) and not isinstance(a, _Sym)]
            trace.append([name, [repr(a) if isinstance(a, _Sym) else a for a in args if isinstance(a, _Sym)][:0], statics])
            outs = tuple(_Sym("%s#%d.%d" % (name, counter[0], k)) for k in range(nout))
            return outs if nout > 1 else outs[0]
        return f

    for nm, nout in (("compute_aabb", 2), ("sort_points_step1", 2), ("sort_points_step2", 4), ("sort_features", 1),
                     ("sort_features_back", 1), ("compute_pdf", 1), ("poisson_sampling", 3),
                     ("get_sampled_features", 1), ("spatial_conv", 1), ("transform_indexs", 1), ("find_neighbors", 2)):
        setattr(mod, nm, rec(nm, nout))
    mod.get_block_size = lambda: 8
    sys.modules["MCConvModule"] = mod

    nu = types.ModuleType("MCNetworkUtils")
    nu.batch_norm_RELU_drop_out = lambda name, f, *a, **k: f
    nu.conv_1x1 = lambda name, f, a, b: f
    nu.MLP_2_hidden = lambda f, *a, **k: f
    nu.MLP_1_hidden = lambda f, *a, **k: f
    sys.modules["MCNetworkUtils"] = nu

    sys.path.insert(0, os.path.join(REF, "utils"))
    sys.path.insert(0, os.path.join(REF, "models"))
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        import MCClassS
        MCClassS.create_network(_Sym("points"), _Sym("batchIds"), _Sym("features"), 1, 32, 16, 40, _Sym("isTraining"),
                                _Sym("kpc"), _Sym("kpf"))
    out = {"model": "MCClassS(numInputFeatures=1, batchSize=32, k=16)", "calls": trace, "variables": variables}
    with open(os.path.join(HERE, "builder_trace.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("builder_trace", len(trace), "records,", len(variables), "variables")


def oracle_chains():
    from oracle.oracle import Oracle
    from tests.helpers import make_cloud, make_mlp, conv_nb, run_chain
    orc = Oracle()
    ident = lambda a: a
    cases = {
        # name: (points source, B, radius, scaleInv, Fin, Fout, combin, poisson radius)
        "cfg0": (("uniform", 2048, 1, 1), 1, 0.1, True, 3, 8, True, 0.1),
        "nonuniform_abs": (("fixture", 0, 0, 0), 1, 0.08, False, 1, 16, True, 0.12),
        "batched_dw": (("clustered", 600, 3, 4), 3, 0.2, True, 8, 8, False, 0.15),
    }
    for name, (src, B, radius, scaleInv, fin, fout, combin, prad) in cases.items():
        if src[0] == "fixture":
            pts = np.load(os.path.join(HERE, "nonuniform_cloud.npz"))["points"][:3000]
            bids = np.zeros((len(pts), 1), np.int32)
        else:
            pts, bids = make_cloud(src[1], src[2], src[3], src[0], src[0] == "clustered")
        feats = (2 * np.random.default_rng(7).random((len(pts), fin)) - 1).astype(np.float32)
        o = run_chain(orc, ident, ident, pts, bids, feats, B, radius, scaleInv, fout=fout, combin=combin,
                      poisson_radius=prad)
        w = o["mlp"]
        outF = fout if combin else fin
        og = (2 * np.random.default_rng(11).random((len(pts), outF)) - 1).astype(np.float32)
        args = (o["sortPts"], o["sortFeatures"], o["sortBatchs"], o["pdfs"], pts, o["startIndexs"], o["packedNeighs"],
                o["aabbMin"], o["aabbMax"], w["w1"], w["w2"], w["w3"], w["b1"], w["b2"], w["b3"])
        conv = orc.spatial_conv(*args, fout, combin, B, radius, scaleInv, True)
        grads = orc.spatial_conv_grad(*args, og, fout, combin, B, radius, scaleInv, True)
        keep = {k: v for k, v in o.items() if isinstance(v, np.ndarray)}
        keep.pop("sortPts"); keep.pop("sortFeatures"); keep.pop("poissonSortPts")
        keep.update(in_points=pts, in_batch_ids=bids, in_features=feats, out_grad=og, conv_out=conv,
                    feat_grad=grads[0], dw1=grads[1], db1=grads[2], dw2=grads[3], db2=grads[4], dw3=grads[5],
                    db3=grads[6], attrs=np.array([B, radius, int(scaleInv), fin, fout, int(combin), prad], np.float64))
        for k, v in w.items():
            keep["mlp_" + k] = v
        np.savez_compressed(os.path.join(HERE, "chain_%s.npz" % name), **keep)
        print("chain", name, "N", len(pts), "E", len(o["packedNeighs"]), "S", len(o["samplePts"]))


if __name__ == "__main__":
    nonuniform_cloud()
    builder_trace()
    oracle_chains()
