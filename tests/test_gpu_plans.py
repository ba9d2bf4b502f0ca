"""Row plans (SELL-64 layouts of a neighbour list, mccnn_rowplan_*) checked slot by slot against a NumPy reconstruction from
the CSR list -- independent of the convolution kernels that consume them. Large lists take the tile fill (forward plan,
records evaluated in the same pass) and the rank + scatter pass of the transposition (transposed plan); small lists the
single-workgroup layout with the gathering fill. What a plan has to hold: every edge of every row exactly once, in the
row's order (forward: list order; transposed: ascending edge id), its record (delta = (p_j - c_i) / R correctly rounded,
1 / (pdf K)) and the index at the other end; every other slot of a slice a zero record and a valid index."""
import numpy as np
import pytest
import torch

from helpers import make_cloud

pytestmark = pytest.mark.gpu


def _decode(plan, rows, e):
    S = plan.num_slices
    base = plan.buf.data_ptr()
    view = lambda addr, count, dt: plan.buf[addr - base: addr - base + count * 4].view(dt).cpu().numpy()
    vrow = view(plan.vrow, 64 * S, torch.int32)
    vcode = view(plan.vcode, 64 * S, torch.int32)
    soff = view(plan.slice_off, S + 1, torch.int32)
    slots = int(soff[S])
    other = view(plan.other, slots, torch.int32)
    rec = view(plan.rec, slots * 4, torch.float32).reshape(slots, 4)
    return vrow, vcode, soff, other, rec


def _check_plan(plan, rows, e, row_start, edge_ids_of_row, other_of_edge, rec_of_edge, other_count):
    vrow, vcode, soff, other, rec = _decode(plan, rows, e)
    S = plan.num_slices
    assert soff[0] == 0 and np.all(np.diff(soff) >= 0) and np.all(np.diff(soff) % 64 == 0)
    deg = np.diff(np.append(row_start[:rows], e))
    seen = np.zeros(e, np.int32)
    real = np.zeros(int(soff[S]), bool)
    pieces = {}
    for s in range(S):
        ln = (soff[s + 1] - soff[s]) // 64
        for lane in range(64):
            r = vrow[s * 64 + lane]
            if r < 0:
                continue
            code = vcode[s * 64 + lane]
            vid = ~code if code < 0 else code
            pieces.setdefault(r, []).append((vid, s, lane, ln, code >= 0))
    L = None
    for r, ps in pieces.items():
        ps.sort()
        assert [p[0] for p in ps] == list(range(ps[0][0], ps[0][0] + len(ps))), "virtual rows of a row are consecutive"
        assert all(p[4] == (len(ps) > 1) for p in ps), "cut flag"
        ids = edge_ids_of_row(r)
        assert len(ids) == deg[r]
        pos = 0
        for k, (vid, s, lane, ln, cut) in enumerate(ps):
            if k == len(ps) - 1:
                n_here = len(ids) - pos
            else:  # a full piece: the longest virtual row there is, so its slice is exactly as long -- the same L everywhere
                n_here = ln
                assert L is None or L == ln
                L = ln
            assert 0 <= n_here <= ln and (L is None or n_here <= L)
            sl = soff[s] + np.arange(n_here) * 64 + lane
            eid = ids[pos:pos + n_here]
            seen[eid] += 1
            real[sl] = True
            assert np.array_equal(other[sl], other_of_edge[eid])
            want = rec_of_edge[eid]
            assert np.array_equal(rec[sl, :3], want[:, :3]), "delta is the correctly rounded quotient"
            assert np.allclose(rec[sl, 3], want[:, 3], rtol=3e-7, atol=0), "1 / (pdf K): v_rcp_f32 is good to 1 ulp"
            pos += n_here
        assert pos == len(ids)
    assert np.all(seen == 1), "every edge exactly once"
    assert set(pieces) == set(range(rows)), "every row has a lane (rows without edges too)"
    pad = ~real
    assert np.all(rec[pad] == 0.0), "padding slots carry zero records"
    assert np.all((other[pad] >= 0) & (other[pad] < other_count)), "... and an index that can be gathered"


@pytest.mark.parametrize("n_per,B,radius", [(9000, 3, 0.08), (1500, 2, 0.12)])
def test_row_plans_slot_by_slot(mc, n_per, B, radius):
    M = mc
    pts, bids = make_cloud(n_per, B, 3, "clustered", True)
    P, Bi = torch.from_numpy(pts).cuda(), torch.from_numpy(bids).cuda()
    F = torch.zeros((len(pts), 8), device="cuda")
    mn, mx = M.compute_aabb(P, Bi, B, True)
    keys, idx = M.sort_points_step1(P, Bi, mn, mx, B, radius, True)
    sP, sB, sF, cells = M.sort_points_step2(P, Bi, F, keys, idx, mn, mx, B, radius, True)
    start, packed = M.find_neighbors(P, Bi, sP, cells, mn, mx, radius, B, True)
    pdfs = M.compute_pdf(sP, sB, mn, mx, start, packed, 0.25, radius, B, True)
    n = m = len(pts)
    e = packed.shape[0]
    args = (sP, sB, pdfs, P, start, packed, mn, mx, n, m, e, B, radius, True, True)
    fwd = M._row_plan(packed, False, *args, centre_points=P)
    tr = M._row_plan(packed, True, *args)
    torch.cuda.synchronize()
    pk = packed.cpu().numpy()
    st = start.cpu().numpy().reshape(-1)
    # records, as the reference spells them (spatial_conv.cu:149-166): delta = (p_j - c_i) / R_b, K = the centre's row length
    sp, sb = sP.cpu().numpy(), sB.cpu().numpy().reshape(-1)
    ext = (mx.cpu().numpy() - mn.cpu().numpy()).max(axis=1).astype(np.float32)
    R = (np.float32(radius) * ext[sb[pk[:, 0]]]).astype(np.float32)
    delta = ((sp[pk[:, 0]] - pts[pk[:, 1]]) / R[:, None]).astype(np.float32)
    K = np.diff(np.append(st, e))[pk[:, 1]].astype(np.float32)
    rec_e = np.concatenate([delta, (np.float32(1) / (pdfs.cpu().numpy().reshape(-1) * K))[:, None]], axis=1).astype(np.float32)
    _check_plan(fwd, m, e, st, lambda r: np.arange(st[r], st[r + 1] if r + 1 < m else e), pk[:, 0], rec_e, n)
    # transposed: rows = neighbour points, edges in ascending edge id
    order = np.argsort(pk[:, 0], kind="stable")
    cnt = np.bincount(pk[:, 0], minlength=n)
    st_t = np.concatenate([[0], np.cumsum(cnt)])[:n]
    start_t, perm_t, _ = packed._mccnn_transposed
    assert np.array_equal(start_t.cpu().numpy()[:n], st_t) and np.array_equal(perm_t.cpu().numpy()[:e], order)
    _check_plan(tr, n, e, st_t, lambda r: order[st_t[r]:st_t[r] + cnt[r]], pk[:, 1], rec_e, m)
