"""Data-parallel equivalence on the GPU: two ranks (gloo, sharing the one GPU of the test box -- the
production backend is nccl == RCCL, one GPU per rank) each take half of a 4-cloud batch with an ABSOLUTE radius, i.e.
the case where the reference uses ONE bounding box for the whole batch (aabb_gpu.cu:104-114). With the box all-reduced
before anything is sorted (PointHierarchy(aabbReduceGroup=...)) every integer output of a shard -- keys, sort order,
cell tables, Poisson samples, neighbour lists -- is the corresponding slice of the single-process batch, and the
all-reduced kernel-MLP gradients equal the single-process gradients within the north-star tolerance."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
B, NPER, R_POISSON, R_CONV, FIN, FOUT = 4, 3000, 0.08, 0.12, 3, 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    from tests.helpers import make_cloud
    pts, bids = make_cloud(NPER, B, 29, "clustered", True)
    rng = np.random.default_rng(31)
    feats = (2 * rng.random((len(pts), FIN)) - 1).astype(np.float32)
    return pts, bids, feats


def _run(pts, bids, feats, batchSize, group):
    """PointHierarchy (one Poisson level) + a same-level convolution and a pooling convolution, backward of a fixed
    linear functional of both outputs. Returns integer records and float results as numpy."""
    import torch
    from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder
    P = torch.from_numpy(pts).cuda()
    Bi = torch.from_numpy(bids).cuda()
    F = torch.from_numpy(feats).cuda().requires_grad_(True)
    ph = PointHierarchy(P, F, Bi, [R_POISSON], "PH", batchSize, False, aabbReduceGroup=group)
    cb = ConvolutionBuilder(KDEWindow=0.2, relativeRadius=False)
    torch.manual_seed(5)  # identical kernel-MLP weights everywhere
    o0 = cb.create_convolution("C0", ph, 0, F, FIN, R_CONV, outNumFeatures=FOUT, multiFeatureConv=True)
    o1 = cb.create_convolution("C1", ph, 0, o0, FOUT, R_CONV, outPointHierarchy=ph, outPointLevel=1)
    loss = (o0 * torch.linspace(-1, 1, FOUT, device="cuda")).sum() + (o1 * torch.linspace(1, 2, FOUT, device="cuda")).sum()
    loss.backward()
    grid = cb.cacheGrids_["PH|0|%s|False" % R_CONV]
    neigh0 = cb.cacheNeighs_["PH|0|%s|False|PH|0" % R_CONV]
    neigh1 = cb.cacheNeighs_["PH|0|%s|False|PH|1" % R_CONV]
    ints = dict(aabbMin=ph.aabbMin_, aabbMax=ph.aabbMax_, samplePts=ph.points_[1], sampleBatchs=ph.batchIds_[1],
                sampleIndexs=ph.sampledIndexs_[0], cellIndexs=grid[2], indexs=grid[3], start0=neigh0[0], packed0=neigh0[1],
                start1=neigh1[0], packed1=neigh1[1])
    ints = {k: v.detach().cpu().numpy() for k, v in ints.items()}
    floats = dict(o0=o0.detach().cpu().numpy(), o1=o1.detach().cpu().numpy(), dF=F.grad.detach().cpu().numpy())
    return ints, floats, cb


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from mccnn_amd.dist import shard_clouds, GradBucket
        pts, bids, feats = _inputs()
        lp, lb, lf, lB, (first, last) = shard_clouds(torch.from_numpy(pts), torch.from_numpy(bids),
                                                     torch.from_numpy(feats), B, rank, world)
        ints, floats, cb = _run(lp.numpy(), lb.numpy().astype(np.int32), lf.numpy(), lB, True)
        GradBucket(cb.parameters()).allreduce()
        grads = {k: v.grad.detach().cpu().numpy() for k, v in cb.named_parameters()}
        q.put((rank, first, last, ints, floats, grads))
    finally:
        dist.destroy_process_group()


def test_two_shards_equal_the_single_process_batch(mc):
    import torch
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    pts, bids, feats = _inputs()
    full_i, full_f, cb = _run(pts, bids, feats, B, None)
    full_g = {k: v.grad.detach().cpu().numpy() for k, v in cb.named_parameters()}
    ids = bids.reshape(-1)
    nc3 = full_i["cellIndexs"].shape[1] ** 3
    sample_cloud = full_i["sampleBatchs"].reshape(-1)
    for rank, first, last, ints, floats, grads in res:
        mask = (ids >= first) & (ids < last)
        p0 = int(np.flatnonzero(mask)[0])                       # points of earlier shards (clouds are contiguous)
        smask = (sample_cloud >= first) & (sample_cloud < last)
        s0 = int(np.flatnonzero(smask)[0])                      # Poisson samples of earlier shards
        # the whole-batch box on every shard
        assert np.array_equal(ints["aabbMin"], full_i["aabbMin"][first:last])
        assert np.array_equal(ints["aabbMax"], full_i["aabbMax"][first:last])
        # grid: sort order and cell table are the shard's slice, shifted by the points that precede it
        assert np.array_equal(ints["indexs"], full_i["indexs"][mask] - p0)
        cells = full_i["cellIndexs"][first:last].copy()
        cells[cells[..., 1] > cells[..., 0]] -= p0
        assert np.array_equal(ints["cellIndexs"], cells) and cells.size == (last - first) * nc3 * 2
        # Poisson level
        assert np.array_equal(ints["samplePts"], full_i["samplePts"][smask])
        assert np.array_equal(ints["sampleBatchs"], full_i["sampleBatchs"][smask] - first)
        assert np.array_equal(ints["sampleIndexs"], full_i["sampleIndexs"][smask] - p0)
        # neighbour lists: same-level (centres = the shard's points) and pooling (centres = its samples)
        for tag, cmask, c0 in (("0", mask, p0), ("1", smask, s0)):
            st = np.append(full_i["start" + tag].reshape(-1), len(full_i["packed" + tag]))
            rows = np.flatnonzero(cmask)
            e0, e1 = st[rows[0]], st[rows[-1] + 1]
            assert np.array_equal(ints["start" + tag].reshape(-1), st[rows] - e0)
            assert np.array_equal(ints["packed" + tag], full_i["packed" + tag][e0:e1] - np.array([p0, c0]))
        # float outputs of the shard: the slice of the single-process result (summation chunking may differ)
        for k, m in (("o0", mask), ("o1", smask), ("dF", mask)):
            ref = full_f[k][m]
            assert np.abs(floats[k] - ref).max() <= 1e-5 * np.abs(full_f[k]).max(), k
        # all-reduced kernel-MLP gradients == single-process gradients
        for k, g in grads.items():
            assert np.abs(g - full_g[k]).max() <= 1e-4 * max(np.abs(full_g[k]).max(), 1e-30), k


def test_bench_through_rccl_with_one_rank(mc):
    """The N > 1 code path of bench.py (RCCL init with device_id, whole-batch box all-reduce, the asynchronous gradient
    all-reduce under the next step's kernels, barriers, max-over-ranks timing) on the one GPU of this box: a process
    group of a single nccl rank (MCCNN_BENCH_FORCE_PG=1). Pipelined steps must survive the all-reduce on RCCL's stream
    and the points/s must be that of the plain run's order of magnitude."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MCCNN_BENCH_FORCE_PG="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "10", "--warmup", "3", "--no-layers",
                        "--no-breakdown", "--no-cpu-baseline", "--no-configs", "--strong-rooms", "2"], env=env, cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    # RCCL may print its version banner on stdout as well: the compact record is the one line that is a JSON object
    # (the LAST line), the full record the earlier `details:` line
    assert r.stdout.strip().splitlines()[-1].startswith("{") and len(r.stdout.strip().splitlines()[-1]) < 4096
    compact, rec = _bench_records(r.stdout)
    assert compact["value"] == rec["value"] and compact["config"]["rccl_world_size"] == 1
    assert rec["config"]["collective_backend"] == "nccl" and rec["n_gpus"] == 1
    assert rec["config"]["pipeline"] is not None and rec["config"]["pipelined_ms_per_step"] is not None
    assert rec["value"] > 5e7
    assert rec["config"]["rccl_world_size"] == 1 and rec["scaling"] == "weak"
    # both curves in one line: the fixed batch (here 2 rooms, all on this rank) beside the one-room-per-rank headline
    assert rec["strong"]["rooms"] == 2 and rec["strong"]["points_total"] == 200000 and rec["strong"]["value"] > 5e7
    assert len(rec["config"]["rank_stats"]["own_ms_per_step"]) == 1


def _bench_records(stdout):
    """(compact record = the last stdout line, full record = the `details:` line before it)."""
    import json
    lines = stdout.strip().splitlines()
    compact = [ln for ln in lines if ln.startswith("{")]
    details = [ln for ln in lines if ln.startswith("details: {")]
    assert len(compact) == 1 and len(details) == 1 and lines[-1] is not None and lines[-1] == compact[0], lines[-3:]
    assert len(compact[0]) < 4096
    return json.loads(compact[0]), json.loads(details[0][len("details: "):])


def _torchrun_bench(nproc, extra, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MCCNN_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MCCNN_BENCH_FORCE_PG"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(nproc), "--steps", "4",
           "--warmup", "2", "--points", "20000", "--no-layers", "--no-breakdown", "--no-cpu-baseline", "--no-configs"] + extra
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    compact, rec = _bench_records(r.stdout)     # rank 0 prints ONE compact record (and its details line)
    assert compact["n_gpus"] == nproc and compact["config"]["rccl_world_size"] == nproc and compact["value"] == rec["value"]
    return rec


def test_bench_eight_ranks_sharing_the_gpu_weak_and_strong(mc):
    """The driver's 8-GPU command line, with eight gloo ranks sharing the one GPU of this box (launch path, whole-batch
    box all-reduce, gradient all-reduce, barriers, max-over-ranks timing, per-rank statistics): one room per rank is both
    the weak-scaling headline and the rank's share of the fixed 8-room batch, so the `strong` object carries the same
    partition."""
    rec = _torchrun_bench(8, [])
    assert rec["n_gpus"] == 8 and rec["config"]["rccl_world_size"] == 8 and rec["scaling"] == "weak"
    # every collective kind ran once before the first warm-up step (a SCALE run times steady state), and a rank's share
    # of the fixed batch IS the N = 1 workload: one room per rank in both curves
    assert rec["config"]["collectives_warmed_before_warmup"] is True
    assert rec["strong"]["rank_stats"] is None or rec["strong"]["rank_stats"]["points"] == [20000] * 8
    assert rec["config"]["points_total"] == 8 * 20000 and rec["config"]["points_per_gpu"] == 20000
    st = rec["config"]["rank_stats"]
    assert len(st["own_ms_per_step"]) == 8 and st["points"] == [20000] * 8 and min(st["own_ms_per_step"]) > 0
    assert rec["strong"]["rooms"] == 8 and rec["strong"]["points_total"] == 8 * 20000 and rec["strong"]["rooms_on_rank0"] == 1
    assert rec["value"] > 0 and rec["strong"]["value"] > 0


def test_bench_two_ranks_strong_batch_of_four_rooms(mc):
    """Strong scaling with several rooms per rank: a fixed batch of 4 rooms over 2 ranks (2 + 2), next to the weak headline
    (1 + 1); and --gpus that disagrees with the launcher is refused instead of silently measuring something else."""
    import subprocess
    import sys
    rec = _torchrun_bench(2, ["--strong-rooms", "4"])
    assert rec["n_gpus"] == 2 and rec["config"]["points_total"] == 2 * 20000
    assert rec["strong"]["rooms"] == 4 and rec["strong"]["rooms_on_rank0"] == 2 and rec["strong"]["points_total"] == 4 * 20000
    assert rec["strong"]["rank_stats"]["points"] == [40000, 40000]
    only = _torchrun_bench(2, ["--scaling", "strong", "--strong-rooms", "4"])
    assert only["scaling"] == "strong" and only["strong"] is None and only["config"]["points_total"] == 4 * 20000
    # a launcher that disagrees with --gpus is refused (one rank started, two asked for) ...
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2"], env=env, cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_bare_bench_with_gpus_2_launches_its_own_ranks(mc):
    """`python bench.py --gpus 2` WITHOUT a launcher (the shape of the driver's N = 1 command with another N): the bench
    re-runs itself under torch.distributed.run -- two ranks sharing this box's GPU over gloo -- and prints one compact
    line with the weak headline AND the strong object."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MCCNN_BENCH_FORCE_PG",
                        "MCCNN_BENCH_BACKEND")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--points", "20000", "--no-layers", "--no-breakdown", "--strong-rooms", "4"], env=env, cwd=root,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    compact, rec = _bench_records(r.stdout)
    assert compact["n_gpus"] == 2 and compact["config"]["rccl_world_size"] == 2 and compact["scaling"] == "weak"
    assert compact["value"] > 0 and compact["strong"]["value"] > 0 and compact["strong"]["rooms"] == 4
    assert compact["config"]["collective_backend"] == "gloo" and compact["config"]["points_total"] == 2 * 20000
