"""The builder counterpart (mccnn_amd.MCConvBuilder) must emit the op sequence, cache behaviour and variable
shapes of the reference's utils/MCConvBuilder.py. The expected trace (tests/golden/builder_trace.json) was recorded
from the REFERENCE code itself running MCClassS's graph builder against recording stubs (tests/golden/make_golden.py).
Here the HIP ops are replaced by oracle-backed CPU shims (test infrastructure) so the test runs without a GPU."""
import json
import math
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture()
def shimmed_builder(oracle, monkeypatch):
    import mccnn_amd.MCConvBuilder as MB
    calls = []
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    n = lambda x: x.detach().numpy() if isinstance(x, torch.Tensor) else x

    def shim(name):
        fn = getattr(oracle, name)

        def f(*args):
            calls.append(name)
            out = fn(*[n(a) for a in args])
            return tuple(t(o) for o in out) if isinstance(out, tuple) else t(out)
        return f

    for nm in ("compute_aabb", "sort_points_step1", "sort_points_step2", "sort_features", "sort_features_back",
               "compute_pdf", "poisson_sampling", "get_sampled_features", "spatial_conv", "transform_indexs",
               "find_neighbors"):
        monkeypatch.setattr(MB, nm, shim(nm))
    monkeypatch.setattr(MB, "get_block_size", lambda: 8)
    return MB, calls


def test_builder_matches_reference_trace(shimmed_builder):
    MB, calls = shimmed_builder
    ref = json.load(open(os.path.join(GOLD, "builder_trace.json")))
    ref_ops = [c[0] for c in ref["calls"] if c[0] not in ("get_variable", "add_to_collection")]
    B, k = 4, 16
    rng = np.random.default_rng(0)
    pts = torch.from_numpy(rng.random((B * 64, 3), dtype=np.float32))
    bids = torch.from_numpy(np.repeat(np.arange(B, dtype=np.int32), 64).reshape(-1, 1))
    feats = torch.ones((B * 64, 1), dtype=torch.float32)
    # models/MCClassS.py:29-71 with the dense layers between the convolutions left out
    ph = MB.PointHierarchy(pts, feats, bids, [0.1, 0.4, math.sqrt(3.0) + 0.1], "MCClassS_PH", B)
    cb = MB.ConvolutionBuilder(KDEWindow=0.2)
    f1 = cb.create_convolution(convName="Conv_1", inPointHierarchy=ph, inPointLevel=0, outPointLevel=1, inFeatures=feats,
                               inNumFeatures=1, outNumFeatures=k, convRadius=0.2, multiFeatureConv=True)
    f1 = torch.cat([f1, f1], 1)  # stands in for conv_1x1(k -> 2k)
    f2 = cb.create_convolution(convName="Conv_2", inPointHierarchy=ph, inPointLevel=1, outPointLevel=2, inFeatures=f1,
                               inNumFeatures=k * 2, convRadius=0.8)
    f2 = torch.cat([f2, f2], 1)
    f3 = cb.create_convolution(convName="Conv_3", inPointHierarchy=ph, inPointLevel=2, outPointLevel=3, inFeatures=f2,
                               inNumFeatures=k * 4, convRadius=math.sqrt(3.0) + 0.1)
    assert calls == ref_ops
    got_vars = {n: list(p.shape) for n, p in cb.named_parameters()}
    assert got_vars == ref["variables"]
    # weight-decay collection: weights, weights2, weights3 of every conv (MCConvBuilder.py:408-416)
    ref_coll = [c[2] for c in ref["calls"] if c[0] == "add_to_collection"]
    assert len(cb.get_collection("weight_decay_loss")) == len(ref_coll) == 9
    assert f3.shape == (ph.points_[3].shape[0], k * 4)
    assert len(ph.points_) == 4 and len(ph.sampledIndexs_) == 3 and ph.radiusList_[0] == 0.0


def test_builder_caches_and_errors(shimmed_builder):
    MB, calls = shimmed_builder
    rng = np.random.default_rng(1)
    pts = torch.from_numpy(rng.random((128, 3), dtype=np.float32))
    bids = torch.zeros((128, 1), dtype=torch.int32)
    feats = torch.from_numpy(rng.random((128, 8), dtype=np.float32))
    ph = MB.PointHierarchy(pts, feats, bids, [], "PH", 1)
    cb = MB.ConvolutionBuilder(KDEWindow=0.2)
    del calls[:]
    a = cb.create_convolution("A", ph, 0, feats, 8, 0.3)
    first = list(calls)
    b = cb.create_convolution("B", ph, 0, a, 8, 0.3)  # same grid / neighbours / pdf -> only sort_features + conv
    assert first == ["sort_points_step1", "sort_points_step2", "find_neighbors", "compute_pdf", "spatial_conv"]
    assert calls[len(first):] == ["sort_features", "spatial_conv"]
    keyGrid = "PH|0|0.3|True"
    assert list(cb.cacheGrids_) == [keyGrid] and list(cb.cacheNeighs_) == [keyGrid + "|PH|0"]
    assert list(cb.cachePDFs_) == [keyGrid + "|PH|0|0.2|True"]
    p_before = dict(cb.named_parameters())
    cb.reset()
    assert not cb.cacheGrids_ and not cb.cacheNeighs_ and not cb.cachePDFs_
    cb.create_convolution("A", ph, 0, feats, 8, 0.3)
    assert all(p_before[k] is v for k, v in cb.named_parameters())  # variables survive reset (get_variable reuse)
    with pytest.raises(RuntimeError):
        cb.create_convolution("C", ph, 0, feats, 8, 0.3, outNumFeatures=4)  # single-feature conv needs Fin == Fout
    ph2 = MB.PointHierarchy(pts, feats, bids, [], "PH2", 2)
    with pytest.raises(RuntimeError):
        cb.create_convolution("D", ph, 0, feats, 8, 0.3, outPointHierarchy=ph2)
    assert b.shape == (128, 8)


def test_prefetch_geometry_is_a_no_op_on_host_tensors(shimmed_builder):
    """The side-stream prefetch exists for the HIP op surface; with host tensors (CPU checker behind the op names) it
    must neither run an op nor park anything, and reset() / create_convolution() behave as before."""
    MB, calls = shimmed_builder
    B = 2
    rng = np.random.default_rng(1)
    pts = torch.from_numpy(rng.random((B * 64, 3), dtype=np.float32))
    bids = torch.from_numpy(np.repeat(np.arange(B, dtype=np.int32), 64).reshape(-1, 1))
    feats = torch.ones((B * 64, 1), dtype=torch.float32)
    ph = MB.PointHierarchy(pts, feats, bids, [], "PH", B)
    cb = MB.ConvolutionBuilder(KDEWindow=0.2)
    before = len(calls)
    cb.prefetch_geometry(ph, 0, 0.3)
    assert len(calls) == before and getattr(cb, "prefetched_", None) is None
    cb.reset()
    assert cb.cacheGrids_ == {} and cb.cacheNeighs_ == {} and cb.cachePDFs_ == {}
    out = cb.create_convolution(convName="Conv", inPointHierarchy=ph, inPointLevel=0, inFeatures=feats, inNumFeatures=1,
                                outNumFeatures=8, convRadius=0.3, multiFeatureConv=True)
    assert out.shape == (B * 64, 8) and len(cb.cacheNeighs_) == 1


def test_running_ahead_is_a_no_op_on_host_tensors(shimmed_builder):
    """PointHierarchy.prefetch() / ConvolutionBuilder.prefetch_step() exist for device tensors and the native executor: with
    host tensors they return None / 0, run no op, and the constructor treats prefetched=None as absent. Attributes whose
    names end in '_' are plain state of the two modules (no parameter / buffer bookkeeping), variables still register."""
    MB, calls = shimmed_builder
    B = 2
    rng = np.random.default_rng(2)
    pts = torch.from_numpy(rng.random((B * 64, 3), dtype=np.float32))
    bids = torch.from_numpy(np.repeat(np.arange(B, dtype=np.int32), 64).reshape(-1, 1))
    feats = torch.ones((B * 64, 1), dtype=torch.float32)
    before = len(calls)
    assert MB.PointHierarchy.prefetch(pts, bids, [0.4], B) is None and len(calls) == before
    ph = MB.PointHierarchy(pts, feats, bids, [0.4], "PH", B, prefetched=None)
    assert len(ph.points_) == 2
    cb = MB.ConvolutionBuilder(KDEWindow=0.2)
    out = cb.create_convolution("Conv", ph, 0, feats, 1, 0.3, outNumFeatures=8, multiFeatureConv=True)
    cb.reset()
    n = len(calls)
    assert cb.prefetch_step(ph) == 0 and len(calls) == n
    assert out.shape == (B * 64, 8)
    cb.hostStepsAhead_ = 0            # (waits for the GPU between steps: nothing to wait for without one)
    cb.reset()
    cb.reset()
    assert "cacheGrids_" in cb.__dict__ and "points_" in ph.__dict__          # plain attributes ...
    assert "Conv_weights" in dict(cb.named_parameters())                        # ... and registered variables


def test_builder_is_a_torch_module_with_reference_variable_names(shimmed_builder):
    """ConvolutionBuilder / PointHierarchy as torch.nn.Modules -- the kernel-MLP variables are
    registered parameters under the reference's names (MCConvBuilder.py:407-419), so parameters(), state_dict(),
    optimisers and parent modules see them; a checkpoint loads into a builder that has not created its variables yet."""
    MB, calls = shimmed_builder
    rng = np.random.default_rng(2)
    pts = torch.from_numpy(rng.random((96, 3), dtype=np.float32))
    bids = torch.zeros((96, 1), dtype=torch.int32)
    feats = torch.from_numpy(rng.random((96, 3), dtype=np.float32))
    ph = MB.PointHierarchy(pts, feats, bids, [], "PH", 1)
    cb = MB.ConvolutionBuilder(KDEWindow=0.2)
    assert isinstance(cb, torch.nn.Module) and isinstance(ph, torch.nn.Module) and len(list(ph.parameters())) == 0
    out = cb(convName="Conv_1", inPointHierarchy=ph, inPointLevel=0, inFeatures=feats, inNumFeatures=3, outNumFeatures=8,
             convRadius=0.3, multiFeatureConv=True)          # forward == create_convolution
    names = [n for n, _ in cb.named_parameters()]
    assert names == ["Conv_1_weights", "Conv_1_biases", "Conv_1_weights2", "Conv_1_biases2", "Conv_1_weights3", "Conv_1_biases3"]
    sd = cb.state_dict()
    assert list(sd) == names and tuple(sd["Conv_1_weights"].shape) == (3, 24) and tuple(sd["Conv_1_weights2"].shape) == (3, 8, 8)
    assert cb.variables_["Conv_1_weights"] is dict(cb.named_parameters())["Conv_1_weights"]
    torch.optim.SGD(cb.parameters(), lr=0.1)                  # an optimiser takes the module's parameters as they are

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.convBuilder = MB.ConvolutionBuilder(KDEWindow=0.2)

    net = Net()                                               # a fresh network: no variables created yet
    net.load_state_dict({"convBuilder." + k: v for k, v in sd.items()})
    assert [n for n, _ in net.named_parameters()] == ["convBuilder." + n for n in names]
    out2 = net.convBuilder.create_convolution(convName="Conv_1", inPointHierarchy=ph, inPointLevel=0, inFeatures=feats,
                                              inNumFeatures=3, outNumFeatures=8, convRadius=0.3, multiFeatureConv=True)
    assert torch.equal(out, out2)                             # the adopted variables are the ones the convolution uses
    with pytest.raises(RuntimeError):                         # tf.get_variable: same name, other shape
        net.convBuilder.create_convolution(convName="Conv_1", inPointHierarchy=ph, inPointLevel=0, inFeatures=feats,
                                           inNumFeatures=3, outNumFeatures=16, convRadius=0.3, multiFeatureConv=True)
