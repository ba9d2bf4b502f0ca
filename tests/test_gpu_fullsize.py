"""BASELINE.json full size (100 000-point non-uniform room, absolute radius 0.1): the oracle is too slow to sit in
the test loop at this size for every op, so parity is checked through size-independent properties of the domain --
sortedness, partition / prefix-sum consistency, symmetry of the neighbour relation, exact radius predicate,
linearity of the convolution in the features, the adjoint identity <conv(F), G> = <F, conv_grad(G)>, a directional
derivative for the weight gradients -- plus an oracle spot check on a subset of centres."""
import numpy as np
import pytest

from tests.helpers import make_room, make_mlp

pytestmark = pytest.mark.gpu
N, R, B = 100000, 0.1, 1


@pytest.fixture(scope="module")
def room(mc):
    import torch
    pts = make_room(N, 20180601)
    P = torch.from_numpy(pts).cuda()
    Bi = torch.zeros((N, 1), dtype=torch.int32, device="cuda")
    F = torch.from_numpy((2 * np.random.default_rng(7).random((N, 1)) - 1).astype(np.float32)).cuda()
    mn, mx = mc.compute_aabb(P, Bi, B, False)
    keys, idx = mc.sort_points_step1(P, Bi, mn, mx, B, R, False)
    sP, sB, sF, cells = mc.sort_points_step2(P, Bi, F, keys, idx, mn, mx, B, R, False)
    start, packed = mc.find_neighbors(P, Bi, sP, cells, mn, mx, R, B, False)
    pdfs = mc.compute_pdf(sP, sB, mn, mx, start, packed, 0.2, R, B, False)
    return dict(pts=pts, P=P, Bi=Bi, F=F, mn=mn, mx=mx, keys=keys, idx=idx, sP=sP, sB=sB, sF=sF, cells=cells,
                start=start, packed=packed, pdfs=pdfs)


def test_grid_invariants(room):
    import torch
    keys, idx, cells = room["keys"], room["idx"].long(), room["cells"].reshape(-1, 2)
    assert torch.equal(torch.sort(idx).values, torch.arange(N, device="cuda"))          # a permutation
    skeys = torch.empty_like(keys)
    skeys[idx] = keys
    assert bool((skeys[1:] >= skeys[:-1]).all())                                          # sorted by cell key
    inv = torch.argsort(idx)
    same = skeys[1:] == skeys[:-1]
    assert bool((inv[1:][same] > inv[:-1][same]).all())                                   # stable inside a cell
    ne = cells[cells[:, 1] > cells[:, 0]]
    assert int((ne[:, 1] - ne[:, 0]).sum()) == N                                          # cells partition [0, N)
    order = torch.argsort(ne[:, 0])
    assert bool((ne[order][1:, 0] == ne[order][:-1, 1]).all())
    assert torch.equal(room["sP"], room["P"][inv]) and torch.equal(room["sF"], room["F"][inv])


def test_neighbor_list_invariants(room, oracle):
    import torch
    start, packed = room["start"][:, 0].long(), room["packed"].long()
    E = packed.shape[0]
    assert int(start[0]) == 0 and bool((start[1:] >= start[:-1]).all()) and int(start[-1]) <= E
    counts = torch.diff(torch.cat([start, torch.tensor([E], device="cuda")]))
    assert torch.equal(packed[:, 1], torch.repeat_interleave(torch.arange(N, device="cuda"), counts))
    # radius predicate in float64 with a 1e-6 band
    d = (room["sP"][packed[:, 0]].double() - room["P"][packed[:, 1]].double()).norm(dim=1)
    assert float(d.max()) < R * (1 + 1e-6)
    # same-level search: the relation is symmetric. (j, i) with i -> its sorted position must map onto itself transposed
    pos = room["idx"].long()                     # original index -> sorted position
    a, b = packed[:, 0], pos[packed[:, 1]]
    fw = torch.sort(a * N + b).values
    bw = torch.sort(b * N + a).values
    assert torch.equal(fw, bw)
    assert bool((counts >= 1).all())             # every point finds at least itself
    # oracle spot check: 300 random centres, exact rows
    rng = np.random.default_rng(0)
    sel = np.sort(rng.choice(N, 300, replace=False))
    st, pk = oracle.find_neighbors(room["pts"][sel], np.zeros((300, 1), np.int32), room["sP"].cpu().numpy(),
                                   room["cells"].cpu().numpy(), room["mn"].cpu().numpy(), room["mx"].cpu().numpy(), R, B,
                                   False)
    pk_gpu, st_gpu = packed.cpu().numpy(), start.cpu().numpy()
    for r, i in enumerate(sel):
        e0, e1 = st[r, 0], (st[r + 1, 0] if r + 1 < 300 else len(pk))
        g0 = st_gpu[i]
        assert np.array_equal(pk_gpu[g0:g0 + (e1 - e0), 0], pk[e0:e1, 0]) and int(counts[i]) == e1 - e0


def test_conv_linearity_adjoint_and_directional_derivative(mc, room):
    import torch
    fin, fout = 1, 64
    w = {k: torch.from_numpy(v).cuda() for k, v in make_mlp(8, 3).items()}
    conv = lambda F, ww=w: mc.spatial_conv(room["sP"], F, room["sB"], room["pdfs"], room["P"], room["start"],
                                           room["packed"], room["mn"], room["mx"], ww["w1"], ww["w2"], ww["w3"], ww["b1"],
                                           ww["b2"], ww["b3"], fout, True, B, R, False, True)
    g = torch.Generator(device="cuda").manual_seed(1)
    F1 = torch.rand((N, fin), device="cuda", generator=g) * 2 - 1
    F2 = torch.rand((N, fin), device="cuda", generator=g) * 2 - 1
    o1, o2, o12 = conv(F1), conv(F2), conv(0.5 * F1 - 2.0 * F2)
    scale = float(o1.abs().max())
    assert float((o12 - (0.5 * o1 - 2.0 * o2)).abs().max()) <= 1e-4 * scale               # linear in the features
    # adjoint identity and weight gradients through autograd
    og = torch.rand((N, fout), device="cuda", generator=g) * 2 - 1
    Fr = F1.clone().requires_grad_(True)
    wr = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    out = conv(Fr, wr)
    out.backward(og)
    lhs = float((out.detach().double() * og.double()).sum())
    rhs = float((F1.double() * Fr.grad.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0), (lhs, rhs)
    # directional derivative along a random direction of all six MLP tensors (central difference, float64 reduction)
    dirs = {k: torch.rand(v.shape, device="cuda", generator=g) * 2 - 1 for k, v in w.items()}
    eps = 1e-3
    lp = float((conv(F1, {k: w[k] + eps * dirs[k] for k in w}).double() * og.double()).sum())
    lm = float((conv(F1, {k: w[k] - eps * dirs[k] for k in w}).double() * og.double()).sum())
    fd = (lp - lm) / (2 * eps)
    an = sum(float((wr[k].grad.double() * dirs[k].double()).sum()) for k in w)
    assert abs(fd - an) <= 2e-2 * max(abs(an), 1.0), (fd, an)                               # ReLU kinks + f32 noise
    # deterministic: bit-identical outputs and parameter gradients across runs
    Fr2 = F1.clone().requires_grad_(True)
    wr2 = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    out2 = conv(Fr2, wr2)
    out2.backward(og)
    assert torch.equal(out, out2)
    for k in w:
        assert torch.equal(wr[k].grad, wr2[k].grad), k


def test_depthwise_full_size(mc, room):
    import torch
    fin = 64
    w = {k: torch.from_numpy(v).cuda() for k, v in make_mlp(8, 5).items()}
    g = torch.Generator(device="cuda").manual_seed(2)
    F = (torch.rand((N, fin), device="cuda", generator=g) * 2 - 1).requires_grad_(True)
    og = torch.rand((N, fin), device="cuda", generator=g) * 2 - 1
    sF = mc.sort_features(F, room["idx"])
    out = mc.spatial_conv(room["sP"], sF, room["sB"], room["pdfs"], room["P"], room["start"], room["packed"], room["mn"],
                          room["mx"], w["w1"], w["w2"], w["w3"], w["b1"], w["b2"], w["b3"], fin, False, B, R, False, True)
    out.backward(og)
    lhs = float((out.detach().double() * og.double()).sum())
    rhs = float((F.detach().double() * F.grad.double()).sum())   # through sort_features' adjoint as well
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0), (lhs, rhs)
    g1 = F.grad.clone()
    F.grad = None
    sF = mc.sort_features(F, room["idx"])
    out2 = mc.spatial_conv(room["sP"], sF, room["sB"], room["pdfs"], room["P"], room["start"], room["packed"], room["mn"],
                           room["mx"], w["w1"], w["w2"], w["w3"], w["b1"], w["b2"], w["b3"], fin, False, B, R, False, True)
    out2.backward(og)
    assert torch.equal(out, out2) and torch.equal(g1, F.grad)   # no atomics anywhere on this path: bit-reproducible


def test_pdf_and_poisson_properties(mc, room):
    import torch
    pdfs = room["pdfs"]
    assert bool((pdfs > 0).all()) and bool(torch.isfinite(pdfs).all())
    sp, sb, si = mc.poisson_sampling(room["sP"], room["sB"], room["cells"], room["mn"], room["mx"], R, B, False)
    S = sp.shape[0]
    assert 0 < S < N and torch.equal(sp, room["sP"][si.long()]) and len(torch.unique(si)) == S
    # separation: no two samples closer than R (checked through a neighbour search among the samples)
    k2, i2 = mc.sort_points_step1(sp, sb, room["mn"], room["mx"], B, R, False)
    p2, b2, _, c2 = mc.sort_points_step2(sp, sb, sp, k2, i2, room["mn"], room["mx"], B, R, False)
    st, pk = mc.find_neighbors(sp, sb, p2, c2, room["mn"], room["mx"], R, B, False)
    assert pk.shape[0] == S                                       # every sample's only neighbour within R is itself
    # maximality: every point has a sample within R
    st2, pk2 = mc.find_neighbors(room["P"], room["Bi"], p2, c2, room["mn"], room["mx"], R, B, False)
    cnt = torch.diff(torch.cat([st2[:, 0], torch.tensor([pk2.shape[0]], device="cuda", dtype=torch.int32)]))
    assert bool((cnt >= 1).all())
