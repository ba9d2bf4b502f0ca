"""N > 1 path on CPU: two gloo ranks. Cloud sharding, the single flat all-reduce of the kernel-MLP gradients and
the (MIN, MAX) all-reduce of the whole-batch bounding box."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mccnn_amd.dist import cloud_partition, shard_clouds, GradBucket, allreduce_aabb


def test_cloud_partition_is_balanced_and_contiguous():
    assert cloud_partition(8, 8) == [(i, i + 1) for i in range(8)]
    assert cloud_partition(16, 4) == [(0, 4), (4, 8), (8, 12), (12, 16)]
    parts = cloud_partition(10, 4)
    assert parts[0][0] == 0 and parts[-1][1] == 10 and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1
    assert cloud_partition(2, 4)[2:] == [(2, 2), (2, 2)]  # more ranks than clouds: empty shards


def test_shard_clouds_rebases_ids():
    B = 5
    bids = torch.tensor(np.repeat(np.arange(B), [3, 1, 4, 2, 5]).reshape(-1, 1).astype(np.int32))
    pts = torch.arange(bids.shape[0] * 3, dtype=torch.float32).reshape(-1, 3)
    feats = torch.arange(bids.shape[0], dtype=torch.float32).reshape(-1, 1)
    seen = 0
    for r in range(2):
        lp, lb, lf, lB, (f, l) = shard_clouds(pts, bids, feats, B, r, 2)
        assert lB == l - f and lb.min().item() == 0 and lb.max().item() == lB - 1
        assert torch.equal(lp, pts[(bids[:, 0] >= f) & (bids[:, 0] < l)])
        seen += lp.shape[0]
    assert seen == pts.shape[0]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        # six MLP tensors of one conv layer with nb = 2 (MCConvBuilder.py:407-419)
        shapes = [(3, 16), (16,), (2, 8, 8), (2, 8), (2, 8, 8), (2, 8)]
        params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
        for k, p in enumerate(params):
            p.grad = torch.full(p.shape, float(rank + 1) * (k + 1))
        params[3].grad = None  # a tensor that received no gradient on this rank
        bucket = GradBucket(params)
        assert bucket.numel == 48 + 16 + 128 + 16 + 128 + 16  # 176 * nb
        bucket.allreduce()
        got = [p.grad.clone() for p in params]
        # asynchronous form (what bench.py uses): a second step's gradients, averaged; values valid after wait()
        for k, p in enumerate(params):
            p.grad = torch.full(p.shape, float(rank + 1) * (k + 1) * 10.0)
        bucket.allreduce(average=True, async_op=True)
        assert bucket.pending is not None
        bucket.wait()
        assert bucket.pending is None
        got += [p.grad.clone() for p in params]
        # a third allreduce() without an explicit wait() first completes the pending one before it repacks
        for k, p in enumerate(params):
            p.grad = torch.full(p.shape, float(rank))
        bucket.allreduce(async_op=True)
        for k, p in enumerate(params):
            p.grad = torch.full(p.shape, 1.0)
        bucket.allreduce()
        got += [p.grad.clone() for p in params]
        # gradients that already lie side by side in one buffer (what spatial_conv's backward hands to autograd) are
        # reduced where they are: no packing, the tensors are not re-pointed
        ext = torch.full((bucket.numel + 5,), float(rank + 1))[5:]  # with a storage offset, as a slice of a larger pool
        off = 0
        for p in params:
            p.grad = ext[off:off + p.numel()].view_as(p)
            off += p.numel()
        before = [p.grad for p in params]
        bucket.allreduce(async_op=True)
        bucket.wait()
        assert all(p.grad is b for p, b in zip(params, before)) and bucket.flat.data_ptr() == ext.data_ptr()
        assert bool((ext == 3.0).all())
        # ... and a buffer that is too short, or out of order, takes the packing path
        params[0].grad, params[1].grad = torch.ones(3, 16), torch.ones(16)
        bucket.allreduce()
        assert bucket.flat is bucket.own and bool((params[0].grad == 2.0).all()) and bool((params[5].grad == 6.0).all())
        mn = torch.tensor([[0.0 + rank, -1.0, 2.0 - rank]])
        mx = torch.tensor([[5.0 + rank, 4.0, 9.0 - rank]])
        mn, mx = allreduce_aabb(mn, mx)
        q.put((rank, [g.numpy() for g in got], mn.numpy(), mx.numpy()))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_allreduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, grads, mn, mx in res:
        assert len(grads) == 18
        for k, g in enumerate(grads[:6]):
            expect = 0.0 if k == 3 else 3.0 * (k + 1)  # (1 + 2) * (k + 1); tensor 3 had no grad anywhere
            assert np.all(g == expect), (rank, k)
        for k, g in enumerate(grads[6:12]):
            assert np.all(g == 15.0 * (k + 1)), (rank, k)  # mean of 10 (k + 1) and 20 (k + 1)
        for k, g in enumerate(grads[12:]):
            assert np.all(g == 2.0), (rank, k)
        assert mn.tolist() == [[0.0, -1.0, 1.0]] and mx.tolist() == [[6.0, 4.0, 9.0]]
