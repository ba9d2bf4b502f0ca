"""one end-to-end model graph on the drop-in path. MCClassS (hierarchy: Poisson sampling, pooling
convolutions between levels, depth-wise convolutions, dense helpers with batch-norm) and MCNormS (two same-level
multi-feature convolutions) run forward + backward through autograd; every parameter must receive a finite gradient,
the gradient must agree with a directional finite difference, and a few optimiser steps must reduce the loss."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def test_mcclass_s_trains(mc):
    import torch
    from mcclass_s import MCClassS, synthetic_batch
    device = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    B, n = 6, 512
    net = MCClassS(1, B, 16, 3, device)
    P, Bi, F, y = synthetic_batch(B, n, 3, rng, device)
    logits = net(P, Bi, F, True, useDropOutFull=False)
    assert logits.shape == (B, 3)  # the last level of MCClassS holds one point per cloud (radius > box diagonal)
    assert [int(p.shape[0]) for p in net.lastHierarchy.points_][-1] == B
    loss = torch.nn.functional.cross_entropy(logits, y)
    loss.backward()
    names = dict(net.convBuilder.named_parameters())
    names.update(dict(net.store.named_parameters()))
    assert len([k for k in names if k.startswith("Conv_")]) == 18      # 3 convs x 6 MLP tensors (MCConvBuilder.py:407-419)
    for k, p in names.items():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
    assert float(names["Conv_1_weights"].grad.abs().sum()) > 0 and float(names["Conv_3_weights3"].grad.abs().sum()) > 0
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    first = None
    for step in range(12):
        logits = net(P, Bi, F, True, useDropOutFull=False)   # same batch: the loss must go down
        loss = torch.nn.functional.cross_entropy(logits, y)
        first = float(loss) if first is None else first
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    assert float(loss) < 0.7 * first, (first, float(loss))
    logits_eval = net(P, Bi, F, False)                         # inference mode uses the moving statistics
    assert bool(torch.isfinite(logits_eval).all())


def test_mcnorm_s_gradient_matches_finite_difference(mc):
    import torch
    from mcclass_s import MCNormS
    device = torch.device("cuda", 0)
    torch.manual_seed(1)
    rng = np.random.default_rng(1)
    B, n = 2, 700
    pts = rng.random((B * n, 3), dtype=np.float32)
    P = torch.from_numpy(pts).to(device)
    Bi = torch.from_numpy(np.repeat(np.arange(B, dtype=np.int32), n).reshape(-1, 1)).to(device)
    F = torch.ones((B * n, 1), device=device)
    net = MCNormS(1, B, 8, device)
    target = torch.from_numpy(rng.normal(size=(B * n, 3)).astype(np.float32)).to(device)
    loss_of = lambda: float(((net(P, Bi, F, True).double() - target.double()) ** 2).mean())
    out = net(P, Bi, F, True)
    loss = ((out - target) ** 2).mean()
    loss.backward()
    params = dict(net.convBuilder.named_parameters())
    g = torch.Generator(device="cuda").manual_seed(3)
    dirs = {k: torch.rand(p.shape, device=device, generator=g) * 2 - 1 for k, p in params.items()}
    an = sum(float((p.grad.double() * dirs[k].double()).sum()) for k, p in params.items())
    eps = 1e-3
    with torch.no_grad():
        for k, p in params.items():
            p.add_(eps * dirs[k])
        lp = loss_of()
        for k, p in params.items():
            p.sub_(2 * eps * dirs[k])
        lm = loss_of()
        for k, p in params.items():
            p.add_(eps * dirs[k])
    fd = (lp - lm) / (2 * eps)
    assert abs(fd - an) <= 3e-2 * max(abs(an), 1e-3), (fd, an)


def test_mcclass_s_cfg1_matches_the_oracle_path(mc):
    """BASELINE cfg1 end to end (MCClassS, 32 clouds x 1 024 points, grow 16): the same graph -- hierarchy, pooling and
    depth-wise MC convolutions, batch-norm / 1x1 / MLP helpers -- once on the HIP ops and once with the CPU oracle behind
    the builder's op names (tests/oracle_ops.py), identical parameters. Logits, loss and the gradient of EVERY parameter
    (18 kernel-MLP tensors, dense layers, batch-norm scales) agree; the dense layers are torch on both sides (GPU vs
    CPU), which is what the tolerances below leave room for."""
    import torch
    from mcclass_s import MCClassS
    from oracle.oracle import Oracle
    from tests.oracle_ops import OracleOps
    from mccnn_amd.workloads import modelnet_like
    B, n, k, ncat = 32, 1024, 16, 40
    pts, bids = modelnet_like(n, B, 51)
    y = np.random.default_rng(2).integers(0, ncat, B)
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    gnet = MCClassS(1, B, k, ncat, dev)
    P, Bi = torch.from_numpy(pts).to(dev), torch.from_numpy(bids).to(dev)
    F = torch.ones((len(pts), 1), device=dev)
    with torch.no_grad():
        gnet(P, Bi, F, True, useDropOutFull=False)              # creates the variables
    cnet = MCClassS(1, B, k, ncat, torch.device("cpu"), ops=OracleOps(Oracle(omp=True)))
    # the models are torch.nn.Modules whose sub-modules register every variable under the reference's name: the whole
    # network state travels through state_dict() / load_state_dict() (variables the CPU twin has not created yet are adopted)
    sd = {k_: v.detach().cpu().clone() for k_, v in gnet.state_dict().items()}
    assert "convBuilder.Conv_1_weights" in sd and "store.Reduce_1_weights" in sd and "store.Reduce_1_In_BN_BN/moving_mean" in sd
    cnet.load_state_dict(sd)
    assert len(list(cnet.parameters())) == len(list(gnet.parameters())) == len(list(gnet.convBuilder.parameters())) + len(
        list(gnet.store.parameters()))
    res = {}
    for tag, net, dv in (("gpu", gnet, dev), ("cpu", cnet, torch.device("cpu"))):
        for p in net.parameters():
            p.grad = None
        logits = net(torch.from_numpy(pts).to(dv), torch.from_numpy(bids).to(dv), torch.ones((len(pts), 1), device=dv), True,
                     useDropOutFull=False)
        loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(y).to(dv))
        loss.backward()
        named = dict(net.convBuilder.named_parameters())
        named.update(dict(net.store.named_parameters()))
        res[tag] = (logits.detach().cpu().numpy(), float(loss), {k_: v.grad.detach().cpu().numpy() for k_, v in named.items()},
                    [int(p.shape[0]) for p in net.lastHierarchy.points_])
    (gl, gloss, gg, gsz), (cl, closs, cg, csz) = res["gpu"], res["cpu"]
    assert gsz == csz and gsz[0] == B * n and gsz[-1] == B
    rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
    # Every tensor is held to the north-star tolerance 1e-4 of its own largest entry. Tensors whose EXACT gradient is zero
    # (biases in front of a batch-norm, which removes the mean: Reduce_*_biases, Final_Logits_biases1/2) hold nothing
    # but the rounding noise of the dense torch layers (GPU vs CPU matmul / batch-norm summation orders, ~1e-8 of the
    # largest gradient in the network), so the scale a difference is measured against is never taken below 1e-3 of
    # the network's largest gradient entry: noise of 1e-7 * gmax passes, a wrong gradient does not.
    gmax = max(float(np.abs(v).max()) for v in cg.values())
    err = {k_: float(np.abs(gg[k_] - cg[k_]).max() / max(np.abs(cg[k_]).max(), 1e-3 * gmax)) for k_ in cg}
    kernel_mlp = {k_: float(np.abs(gg[k_] - cg[k_]).max() / max(np.abs(cg[k_]).max(), 1e-30)) for k_ in cg if k_.startswith("Conv_")}
    assert max(kernel_mlp.values()) <= 1e-4, kernel_mlp   # the hot path's own 18 tensors: against their own scale, no floor
    worst = max(err, key=err.get)
    print("cfg1 end to end: logits %.1e loss %.1e worst gradient %.1e (%s) over %d tensors" % (
        rel(gl, cl), abs(gloss - closs), err[worst], worst, len(cg)))
    assert rel(gl, cl) <= 1e-4 and abs(gloss - closs) <= 1e-5 * abs(closs)
    assert set(gg) == set(cg) and len([k_ for k_ in cg if k_.startswith("Conv_")]) == 18
    for k_ in cg:
        assert err[k_] <= 1e-4, (k_, err[k_])
