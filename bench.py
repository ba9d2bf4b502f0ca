#!/usr/bin/env python3
"""bench.py -- MC-convolved points/sec (fwd+bwd) on a 100k-point non-uniform room, radius 0.1.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one forward + backward Monte-Carlo convolution over one batch, driven through the
drop-in builder API (mccnn_amd.MCConvBuilder.ConvolutionBuilder.create_convolution):
    fwd = sort_points_step1, sort_points_step2, find_neighbors, compute_pdf, spatial_conv
    bwd = spatial_conv_grad, sort_points_step2_grad            
plus, for N > 1, ONE RCCL all-reduce of the flattened kernel-MLP weight gradients. The batch shards
cloud-per-GPU (one 100k-point room per rank, weak scaling); there is no data-path collective.
Inputs are synthetic and resident in HBM before the timed region. The timed region is bracketed by
barrier + torch.cuda.synchronize() on both sides and the max over ranks is reported.

Rank 0 prints ONE JSON line (see README / DESIGN.md for the field contract), including
  roofline     -- the dominant kernel's algorithmic flops (or bytes) / its HIP-event duration
  cpu_baseline -- the CPU oracle (OpenMP port of the reference algorithms; the reference's own ops
                  are GPU-only and cannot run on a CPU) timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LAYERS = {  # name: (Fin, Fout, combin)   -- SURVEY 8(d) layer shapes
    "1to64": (1, 64, True),     # MCSegScanNet first layer at grow 64 (models/MCSegScanNet.py:37-47)
    "3to8": (3, 8, True),       # cfg0 shape
    "dw256": (256, 256, False),  # wide depth-wise layer (Up_0_1 at grow 64)
}
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8 TB/s
F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32 vector == f32 MFMA dense peak


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_inputs(npts, rooms, rank, layer, device):
    from tests.helpers import make_room
    fin, fout, combin = LAYERS[layer]
    pts = np.concatenate([make_room(npts, 20180601 + rank * rooms + r) for r in range(rooms)])
    bids = np.repeat(np.arange(rooms, dtype=np.int32), npts).reshape(-1, 1)
    rng = np.random.default_rng(7 + rank)
    feats = (2 * rng.random((len(pts), fin)) - 1).astype(np.float32)
    outF = fout if combin else fin
    ograd = (2 * np.random.default_rng(11 + rank).random((len(pts), outF)) - 1).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(device)
    return pts, bids, feats, ograd, t(pts), t(bids), t(feats), t(ograd)


def ev_time(fn, iters=5):
    """Average HIP-event duration (ms) of fn() on the current stream, queue drained before each call."""
    ts = []
    for _ in range(iters):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.mean(ts)), r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=100000, help="points per room")
    ap.add_argument("--rooms-per-gpu", type=int, default=1)
    ap.add_argument("--layer", choices=sorted(LAYERS), default="1to64")
    ap.add_argument("--radius", type=float, default=0.1)
    ap.add_argument("--window", type=float, default=0.2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # one process per GPU; MCCNN_BENCH_BACKEND=gloo lets several ranks share one GPU (single-GPU smoke test of the
    # N > 1 code path only -- the real runs use nccl == RCCL over xGMI)
    backend = os.environ.get("MCCNN_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    if world != args.gpus:
        log("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world))

    from mccnn_amd import build as mbuild
    if rank == 0 and mbuild.needs_build():
        mbuild.build()
    if world > 1:
        dist.barrier()
    from mccnn_amd import MCConvModule as M
    from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder
    from mccnn_amd.dist import GradBucket, allreduce_aabb

    fin, fout, combin = LAYERS[args.layer]
    B = args.rooms_per_gpu
    npts = args.points
    pts_np, bids_np, feats_np, ograd_np, P, Bi, F, OG = make_inputs(npts, B, rank, args.layer, device)
    F.requires_grad_(True)
    m_local = P.shape[0]

    # level-0-only hierarchy: computes the (whole-batch, absolute-radius) bounding box once, outside the step
    ph = PointHierarchy(P, F, Bi, [], "bench_PH", B, False)
    if world > 1:
        allreduce_aabb(ph.aabbMin_, ph.aabbMax_)
    builder = ConvolutionBuilder(KDEWindow=args.window, relativeRadius=False)
    torch.manual_seed(1234)  # identical kernel-MLP weights on every rank

    state = {"bucket": None}

    def step():
        builder.reset()
        F.grad = None
        for p in builder.parameters():
            p.grad = None
        out = builder.create_convolution("Conv", ph, 0, F, fin, args.radius, outNumFeatures=fout,
                                         multiFeatureConv=combin, KDEWindow=args.window)
        out.backward(OG)
        if world > 1:
            if state["bucket"] is None:  # the variables exist after the first create_convolution
                state["bucket"] = GradBucket(builder.parameters())
            state["bucket"].allreduce()
        return out

    out = step()  # creates the variables
    e_local = int(next(iter(builder.cacheNeighs_.values()))[1].shape[0])
    for _ in range(max(args.warmup - 1, 0)):
        step()

    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        cnt = torch.tensor([m_local], dtype=torch.float64, device=device)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        m_total = float(cnt.item())
    else:
        m_total = float(m_local)
    ms_per_step = elapsed / args.steps * 1e3
    value = m_total * args.steps / elapsed

    # ------------------------------------------------------------------ per-op breakdown + roofline (rank 0)
    roofline, breakdown = None, None
    if rank == 0 and not args.no_breakdown:
        mn, mx = ph.aabbMin_, ph.aabbMax_
        r, w = args.radius, args.window
        Fd = F.detach()
        t_s1, (keys, idx) = ev_time(lambda: M.sort_points_step1(P, Bi, mn, mx, B, r, False))
        t_s2, (sP, sB, sF, cells) = ev_time(lambda: M.sort_points_step2(P, Bi, Fd, keys, idx, mn, mx, B, r, False))
        t_fn, (start, packed) = ev_time(lambda: M.find_neighbors(P, Bi, sP, cells, mn, mx, r, B, False))
        t_pdf, pdfs = ev_time(lambda: M.compute_pdf(sP, sB, mn, mx, start, packed, w, r, B, False))
        ws = [p.detach() for p in builder.parameters()]  # weights, biases, weights2, biases2, weights3, biases3
        nb = ws[0].shape[1] // 8
        w1, b1, w2, b2, w3, b3 = ws[0], ws[1], ws[2].reshape(8, -1), ws[3].reshape(-1), ws[4].reshape(8, -1), ws[5].reshape(-1)
        lib = M._lib.load()
        from mccnn_amd._lib import ptr, stream_handle, check
        n, m, e = sP.shape[0], P.shape[0], packed.shape[0]
        outF = fout if combin else fin
        o = torch.empty((m, outF), dtype=torch.float32, device=device)
        fwd_ws = torch.empty(max(256, lib.mccnn_spatial_conv_fwd_workspace_bytes(m, e, fin, fout, int(combin))),
                             dtype=torch.uint8, device=device)
        conv_args = (ptr(sP), ptr(sF), ptr(sB), ptr(pdfs), ptr(P), ptr(start), ptr(packed), ptr(mn), ptr(mx), ptr(w1),
                     ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3))
        sbytes = lib.mccnn_spatial_conv_state_bytes(m, fin, fout, int(combin))
        state = torch.empty(sbytes, dtype=torch.uint8, device=device) if sbytes else None  # kept fwd -> bwd, as autograd does
        t_fwd, _ = ev_time(lambda: check(lib.mccnn_spatial_conv_fwd(*conv_args, n, m, e, fin, fout, int(combin), B, r, 0,
                                                                    1, ptr(o), ptr(state), ptr(fwd_ws), fwd_ws.numel(),
                                                                    stream_handle()), "conv_fwd"))
        fg = torch.empty_like(sF)
        gws = [torch.empty_like(t) for t in (w1, b1, w2, b2, w3, b3)]
        bwd_ws = torch.empty(max(256, lib.mccnn_spatial_conv_bwd_workspace_bytes(n, m, e, fin, fout, int(combin))),
                             dtype=torch.uint8, device=device)
        # start_t / perm_t = NULL: the call builds the transposed list itself (worst case: no sharing across layers)
        t_bwd, _ = ev_time(lambda: check(lib.mccnn_spatial_conv_bwd(*conv_args, ptr(OG), n, m, e, fin, fout, int(combin), B,
                                                                    r, 0, 1, ptr(state), None, None, ptr(fg),
                                                                    *[ptr(g) for g in gws],
                                                                    ptr(bwd_ws), bwd_ws.numel(), stream_handle()),
                                         "conv_bwd"))
        t_s2g, _ = ev_time(lambda: M._gather_rows(fg, idx, n))
        C = int(np.prod(cells.shape[:4]))
        # algorithmic work per launch (stated in DESIGN.md)
        alg = {
            "sort_points_step1": ("hbm", n * 16 + n * 8 + 4 * C, t_s1),
            "sort_points_step2": ("hbm", n * (16 + 8 + 4 * fin) + n * (16 + 4 * fin) + 8 * C, t_s2),
            "find_neighbors": ("hbm", 16 * m + 12 * n + 8 * C + 4 * m + 8 * e, t_fn),
            "compute_pdf": ("hbm", 24 * e, t_pdf),
            "spatial_conv_fwd": ("mfma", 320.0 * nb * e, t_fwd),
            "spatial_conv_bwd": ("mfma", 912.0 * nb * e, t_bwd),
            "sort_points_step2_grad": ("hbm", n * (4 + 8 * fin), t_s2g),
        }
        breakdown = {}
        for k, (bound, work, ms) in alg.items():
            ach = work / (ms * 1e-3) / (1e9 if bound == "hbm" else 1e12)
            breakdown[k] = {"ms": round(ms, 4), "bound": bound, "achieved": round(ach, 3),
                            "unit": "GB/s" if bound == "hbm" else "TFLOP/s"}
        dom = max(alg, key=lambda k: alg[k][2])
        bound, work, ms = alg[dom]
        peak = HBM_PEAK_GBS if bound == "hbm" else F32_PEAK_TFLOPS
        ach = work / (ms * 1e-3) / (1e9 if bound == "hbm" else 1e12)
        roofline = {"kernel": dom, "bound": bound, "achieved": round(ach, 3), "peak": peak,
                    "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                    "ms": round(ms, 4), "edges": e, "mlp_blocks": nb}
        if combin and fin == 1:
            # one-input-feature layers run the factored kernels (conv_f1.hip): layer 3 is applied per centre, so fewer
            # flops are EXECUTED than the algorithm of SURVEY 8d counts; `achieved` keeps the contract's algorithmic
            # figure (an effective rate), this is the rate of the arithmetic actually issued
            ex = {"spatial_conv_fwd": 192.0 * nb * e + 144.0 * nb * m, "spatial_conv_bwd": 528.0 * nb * e + 290.0 * nb * m}
            if dom in ex:
                roofline["executed_tflops"] = round(ex[dom] / (ms * 1e-3) / 1e12, 3)
                roofline["executed_frac"] = round(roofline["executed_tflops"] / peak, 4)
        # HBM-side bytes per launch of the dominant kernel come from the separate rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE passes of THIS command (tools/prof.sh), committed under profiles/ -- bench.py cannot collect
        # counters itself; null when the committed profile does not cover this workload.
        tfile = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic_%s.json" % args.layer)
        if dom == "spatial_conv_bwd" and os.path.exists(tfile) and args.points == 100000 and args.rooms_per_gpu == 1:
            with open(tfile) as fh:
                for kname, tr in json.load(fh).items():
                    if kname.startswith("conv_bwd_mfma") or kname.startswith("f1_bwd_edges"):
                        roofline["traffic"] = int(tr["bytes"])
                        roofline["traffic_source"] = "profiles/" + os.path.basename(tfile)

    # ------------------------------------------------------------------ CPU baseline (rank 0, N == 1)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            from oracle.oracle import Oracle
            orc = Oracle(omp=True)
            ws = [p.detach().cpu().numpy() for p in builder.parameters()]
            w1, b1, w2, b2, w3, b3 = ws[0], ws[1], ws[2].reshape(8, -1), ws[3].reshape(-1), ws[4].reshape(8, -1), ws[5].reshape(-1)
            mn, mx = orc.compute_aabb(pts_np, bids_np, B, False)
            def cpu_step():
                k, i = orc.sort_points_step1(pts_np, bids_np, mn, mx, B, args.radius, False)
                sp, sb, sf, cl = orc.sort_points_step2(pts_np, bids_np, feats_np, k, i, mn, mx, B, args.radius, False)
                st, pk = orc.find_neighbors(pts_np, bids_np, sp, cl, mn, mx, args.radius, B, False)
                pdf = orc.compute_pdf(sp, sb, mn, mx, st, pk, args.window, args.radius, B, False)
                a = (sp, sf, sb, pdf, pts_np, st, pk, mn, mx, w1, w2, w3, b1, b2, b3)
                oc = orc.spatial_conv(*a, fout, combin, B, args.radius, False, True)
                g = orc.spatial_conv_grad(*a, ograd_np, fout, combin, B, args.radius, False, True)
                orc.sort_points_step2_grad(i, np.zeros_like(sp), g[0])
                return oc

            # bounded sample: whole steps of the identical workload until ~3 s of wall time (>= 2 steps), best step
            times = []
            tstart = time.perf_counter()
            while len(times) < 2 or (time.perf_counter() - tstart < 3.0 and len(times) < 8):
                c0 = time.perf_counter()
                oc = cpu_step()
                times.append(time.perf_counter() - c0)
            best = min(times)
            cpu = {"value": round(m_local / best, 1), "unit": "points/s", "cores": orc.num_threads(),
                   "kind": "port",
                   "sample": "%d steps (fwd+bwd) of the identical workload, OpenMP over centres, best step %.2f s "
                             "(%.0f core-seconds in total)" % (len(times), best, sum(times) * orc.num_threads())}
            # cross-check the GPU result of the timed workload against the oracle (same inputs)
            err = float(np.abs(out.detach().cpu().numpy() - oc).max() / max(np.abs(oc).max(), 1e-30))
            cpu["gpu_vs_oracle_max_rel_err"] = float("%.3e" % err)
        except Exception as ex:  # the baseline is informative; never fail the bench on it
            cpu = {"error": repr(ex)}

    if rank == 0:
        rec = {
            "metric": "MC-convolved points/sec (fwd+bwd), 100k-pt cloud r=0.1",
            "value": round(value, 1), "unit": "points/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ScanNet-like non-uniform room, %d pts/room, %d room(s)/GPU, absolute radius %g, "
                                   "KDE window %g, same-level conv %s (Fin=%d, Fout=%d, %s), avg on"
                                   % (npts, B, args.radius, args.window, args.layer, fin, fout,
                                      "combin" if combin else "depth-wise"),
                       "points_per_gpu": m_local, "edges_per_gpu": e_local, "layer": args.layer,
                       "parallelism": "cloud-per-GPU dp%d" % world},
            "roofline": roofline, "cpu_baseline": cpu, "breakdown": breakdown,
        }
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
