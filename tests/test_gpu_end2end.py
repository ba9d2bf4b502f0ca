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
