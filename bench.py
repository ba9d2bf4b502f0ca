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
cloud-per-GPU; there is no data-path collective.
    headline (`value`, `scaling: weak`): one 100k-point room per rank -- BASELINE cfg4 at N = 8
    `strong` object: a fixed batch of --strong-rooms (8) rooms split over the N ranks (the north star's "strong scaling
    on batched clouds"); measured in the same run, so the driver's per-N lines carry both curves.
    --scaling weak|strong restricts the run to one of them (strong: the fixed batch becomes the headline).
Inputs are synthetic and resident in HBM before the timed region. The timed region is bracketed by
barrier + torch.cuda.synchronize() on both sides and the max over ranks is reported.

Steps are PIPELINED by default (--no-pipeline for strictly sequential steps): the geometry ops of a batch (grid build,
neighbour search, KDE -- they depend on the points only) are launched on a side stream while the convolution kernels of
the previous batch run (ConvolutionBuilder.prefetch_geometry). Every step still executes every op: inside the timed
region K geometry passes and K convolution passes run. Before timing, three pipelined steps are checked to reproduce a
sequential step's forward output bit for bit; config.sequential_ms_per_step reports the same K steps unpipelined.

Rank 0 prints ONE JSON line (see README / DESIGN.md for the field contract), including
  roofline     -- the dominant kernel's algorithmic flops (or bytes) / its HIP-event duration
  layers       -- the same measurement for all three layer shapes of SURVEY 8d (1to64, 3to8, dw256), so that the
                  headline shape cannot hide the depth-wise regime
  configs      -- BASELINE.json configs cfg0..cfg4 (mccnn_amd.workloads): one step = PointHierarchy + forward +
                  backward of EVERY convolution of the model's graph; points/s, launches per step, per-layer conv
                  times with their rooflines, and the CPU port on a bounded sample of the same graph
                  (--config cfgN: only that one; --no-configs: none)
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
PROFILE_ROUND = "r06"    # profiles/<round>_pmc_traffic_<layer>.json supplies roofline.traffic


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def conv_bound(fin, combin):
    """combin layers and small nb are bounded by matrix/FMA throughput, wide depth-wise layers by the
    gather of feature / gradient rows (4 Fin bytes per edge and direction)."""
    return "hbm" if (not combin and fin >= 64) else "mfma"


def conv_work(which, fin, fout, combin, nb, n, m, e):
    """(bound, algorithmic work per launch) of spatial_conv forward / backward."""
    outF = fout if combin else fin
    if conv_bound(fin, combin) == "mfma":
        return "mfma", (320.0 if which == "fwd" else 912.0) * nb * e
    fwd_bytes = e * (24 + 4 * fin) + m * (16 + 4 * outF)
    if which == "fwd":
        return "hbm", float(fwd_bytes)
    # backward re-reads what forward reads, gathers the out-gradient row per edge, adds the E*4Fin feature-gradient
    # contribution, reads M*4Fout and writes the 704*nb parameter gradients
    return "hbm", float(fwd_bytes + e * 4 * fin + e * 4 * fin + m * 4 * outF + 704 * nb)


def make_inputs(npts, seeds, layer, device, rank):
    from mccnn_amd.workloads import rooms
    fin, fout, combin = LAYERS[layer]
    pts, bids = rooms(npts, seeds)
    rng = np.random.default_rng(7 + rank)
    feats = (2 * rng.random((len(pts), fin)) - 1).astype(np.float32)
    outF = fout if combin else fin
    ograd = (2 * np.random.default_rng(11 + rank).random((len(pts), outF)) - 1).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(device)
    return pts, bids, feats, ograd, t(pts), t(bids), t(feats), t(ograd)


def ev_time(fn, iters=20):
    """Average HIP-event duration (ms) of fn() on the current stream (the one every C-ABI call is launched on): two untimed
    calls, then `iters` calls back to back between one pair of events -- the duration of a launch in a loop, as in the
    step loops (a call timed alone on a drained queue runs on a chip that has just idled: +1-4 % on the 0.37 ms backward
    pass, and noisier)."""
    r = fn()
    r = fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        r = fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters, r


class Workload:
    """One layer shape on this rank's share of the batch."""

    def __init__(self, args, layer, seeds, rank, world, device):
        from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder
        self.args, self.layer, self.world, self.device = args, layer, world, device
        self.dist_on = dist.is_available() and dist.is_initialized()
        self.fin, self.fout, self.combin = LAYERS[layer]
        self.B = len(seeds)
        (self.pts_np, self.bids_np, self.feats_np, self.ograd_np, self.P, self.Bi, self.F, self.OG) = make_inputs(
            args.points, seeds, layer, device, rank)
        self.F.requires_grad_(True)
        # level-0-only hierarchy: computes the (whole-batch, absolute-radius) bounding box once, outside the step; with
        # N > 1 the shards all-reduce it before anything is sorted (aabb_gpu.cu:104-114)
        self.ph = PointHierarchy(self.P, self.F, self.Bi, [], "bench_PH", self.B, False,
                                 aabbReduceGroup=True if self.dist_on else None)
        self.builder = ConvolutionBuilder(KDEWindow=args.window, relativeRadius=False)
        torch.manual_seed(1234)  # identical kernel-MLP weights on every rank
        self.bucket = None
        self.params = []
        self.pipeline = False
        self.skip_allreduce = False
        from mccnn_amd import _env as _menv, native as _native
        self.native_prefetch = bool(self.builder.native_ and _native.side_streams_available()
                                    and _menv.debug("native_prefetch", True))
        # what the prefetch also starts for the backward pass: depth-wise layers sweep the transposed row plan. Combin layers
        # with 2..4 input features CAN gather their feature gradient through the transposed list (no float atomics,
        # bit-reproducible: MCCNN_PF_TLIST=1) -- measured on 3to8: pipelined step 0.590 -> 0.739 ms (the transposition's
        # atomics run beside the convolutions), so the scatter stays the default of a single layer
        self.pf_transposed = True if not self.combin else ("list" if 2 <= self.fin <= 4 and os.environ.get(
            "MCCNN_PF_TLIST", "0") == "1" else False)
        self.out = self.step()  # creates the variables (strictly sequential step)
        self.params = list(self.builder.parameters())
        ok = not getattr(args, "no_pipeline", False)
        if ok:
            # pipelined steps must reproduce the sequential forward output bit for bit (the forward is deterministic);
            # anything else -- a mismatch or an exception on ANY rank -- switches the pipeline off for this run. No
            # collective runs inside the check, so a rank that fails cannot leave the others waiting.
            self.skip_allreduce = True
            try:
                self.pipeline = True
                for _ in range(3):
                    ok = ok and bool(torch.equal(self.step(), self.out))
                torch.cuda.synchronize()
            except Exception as ex:
                log("pipeline check failed: %r" % (ex,))
                ok = False
            self.skip_allreduce = False
            self.pipeline = False
            self.builder.prefetched_ = None
            self.builder.prefetchedGeo_ = {}
            torch.cuda.synchronize()
        if self.dist_on:
            flag = torch.tensor([1.0 if ok else 0.0], device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item() > 0.5)
        if not ok and not getattr(args, "no_pipeline", False):
            log("pipeline disabled")
        self.pipeline = ok
        self.e_local = int(next(iter(self.builder.cacheNeighs_.values()))[1].shape[0])

    def step(self):
        from mccnn_amd.dist import GradBucket
        a = self.args
        self.builder.reset()
        self.F.grad = None
        for p in self.params:
            p.grad = None
        early = self.pipeline and self.native_prefetch
        mid = early and os.environ.get("MCCNN_PF_PLACE", "0") == "1"  # (A/B: between forward and backward -- measured slower)
        if early and not mid:
            # geometry of the NEXT batch (grid build, neighbour search, KDE: it depends on the points only) on a side
            # stream, under the convolution kernels of THIS batch; the next reset() installs it. On the native path the
            # side stream forks behind what the calling stream holds at the call, so the call comes before this batch's
            # convolutions are launched (ConvolutionBuilder.__prefetch_native__)
            self.builder.prefetch_geometry(self.ph, 0, a.radius, KDEWindow=a.window, transposed=self.pf_transposed)
        out = self.builder.create_convolution("Conv", self.ph, 0, self.F, self.fin, a.radius, outNumFeatures=self.fout,
                                              multiFeatureConv=self.combin, KDEWindow=a.window)
        if mid:
            self.builder.prefetch_geometry(self.ph, 0, a.radius, KDEWindow=a.window, transposed=self.pf_transposed)
        out.backward(self.OG)
        if self.pipeline and not early:
            # (op-by-op prefetch, MCCNN_NATIVE=0 / no torch extension: its side stream waits for the hierarchy's and the
            # last reset()'s events only, and its host work is better spent after this batch's launches)
            self.builder.prefetch_geometry(self.ph, 0, a.radius, KDEWindow=a.window, transposed=self.pf_transposed)
        if self.dist_on and not self.skip_allreduce:
            if self.bucket is None:  # the variables exist after the first create_convolution
                self.bucket = GradBucket(list(self.builder.parameters()), single_rank=True)
            # enqueued on RCCL's stream; the next step's grid build, search and forward pass run under it (the reduced
            # gradients are not read before the next pack, or the wait() that closes the timed region)
            self.bucket.allreduce(async_op=True)
        return out

    def timed(self, steps, warmup):
        """K timed steps after W warm-up steps; barrier + synchronize on both sides; max over ranks. self.rank_stats holds
        every rank's own loop time and how long it then waited for the last gradient all-reduce (a slow rank shows)."""
        for _ in range(max(warmup, 0)):
            self.step()
        if self.bucket is not None:
            self.bucket.wait()
        torch.cuda.synchronize()
        if self.dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        trace = [] if os.environ.get("MCCNN_BENCH_TRACE") else None  # diagnostic: per-step times of the region to stderr
        hlag = int(os.environ.get("MCCNN_BENCH_LAG_HEADLINE", "0"))  # A/B: the host at most `hlag` steps ahead of the GPU
        hevs = []
        for _ in range(steps):
            if hlag and len(hevs) >= hlag:
                hevs[-hlag].synchronize()
            self.step()
            if hlag:
                e_ = torch.cuda.Event()
                e_.record()
                hevs.append(e_)
            if trace is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                trace.append((ev, time.perf_counter() - t0))
        torch.cuda.current_stream().synchronize()  # this rank's own kernels
        t_own = time.perf_counter() - t0
        if trace:
            log("per-step GPU ms: " + " ".join("%.3f" % trace[i][0].elapsed_time(trace[i + 1][0]) for i in range(len(trace) - 1))
                + " | host issue ms: " + " ".join("%.3f" % ((trace[i + 1][1] - trace[i][1]) * 1e3) for i in range(len(trace) - 1)))
        if self.bucket is not None:
            self.bucket.wait()  # the last step's all-reduce finishes inside the timed region
        torch.cuda.synchronize()
        t_wait = time.perf_counter() - t0 - t_own
        if self.dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        m_local = self.P.shape[0]
        self.rank_stats = None
        if self.dist_on:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=self.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
            cnt = torch.tensor([m_local], dtype=torch.float64, device=self.device)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
            m_total = float(cnt.item())
            mine = torch.tensor([t_own / steps * 1e3, t_wait * 1e3, float(m_local)], dtype=torch.float64, device=self.device)
            every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(every, mine)
            self.rank_stats = {"own_ms_per_step": [round(float(t[0]), 4) for t in every],
                               "allreduce_wait_ms_at_region_end": [round(float(t[1]), 4) for t in every],
                               "points": [int(t[2]) for t in every]}
        else:
            m_total = float(m_local)
        return elapsed / steps * 1e3, m_total * steps / elapsed, m_total

    def find_neighbors_roofline(self):
        """find_neighbors alone on THIS workload's batch (the north star's gather-bound pass): HIP-event time of the op
        through the op surface, algorithmic bytes 16M + 12N + 8C + 4M + 8E of SURVEY 8(d) against the HBM peak."""
        from mccnn_amd import MCConvModule as M
        a, B, P, Bi = self.args, self.B, self.P, self.Bi
        mn, mx = self.ph.aabbMin_, self.ph.aabbMax_
        keys, idx = M.sort_points_step1(P, Bi, mn, mx, B, a.radius, False)
        sP, sB, sF, cells = M.sort_points_step2(P, Bi, self.F.detach(), keys, idx, mn, mx, B, a.radius, False)
        t_fn, (start, packed) = ev_time(lambda: M.find_neighbors(P, Bi, sP, cells, mn, mx, a.radius, B, False))
        n, m, e = sP.shape[0], P.shape[0], packed.shape[0]
        C = int(np.prod(cells.shape[:4]))
        work = 16 * m + 12 * n + 8 * C + 4 * m + 8 * e
        ach = work / (t_fn * 1e-3) / 1e9
        return {"bound": "hbm", "ms": round(t_fn, 4), "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "rooms": B, "points": int(n), "edges": int(e),
                "algorithmic_bytes": int(work)}

    def breakdown(self):
        """Per-op HIP-event times through the C-ABI on rank 0 and the roofline of the dominant op."""
        from mccnn_amd import MCConvModule as M
        from mccnn_amd._lib import ptr, stream_handle, check
        a, B, P, Bi, device = self.args, self.B, self.P, self.Bi, self.device
        fin, fout, combin = self.fin, self.fout, self.combin
        mn, mx = self.ph.aabbMin_, self.ph.aabbMax_
        r, w = a.radius, a.window
        Fd = self.F.detach()
        t_s1, (keys, idx) = ev_time(lambda: M.sort_points_step1(P, Bi, mn, mx, B, r, False))
        t_s2, (sP, sB, sF, cells) = ev_time(lambda: M.sort_points_step2(P, Bi, Fd, keys, idx, mn, mx, B, r, False))
        t_fn, (start, packed) = ev_time(lambda: M.find_neighbors(P, Bi, sP, cells, mn, mx, r, B, False))
        t_pdf, pdfs = ev_time(lambda: M.compute_pdf(sP, sB, mn, mx, start, packed, w, r, B, False))
        ws = [p.detach() for p in self.builder.parameters()]  # weights, biases, weights2, biases2, weights3, biases3
        nb = ws[0].shape[1] // 8
        w1, b1, w2, b2, w3, b3 = ws[0], ws[1], ws[2].reshape(8, -1), ws[3].reshape(-1), ws[4].reshape(8, -1), ws[5].reshape(-1)
        lib = M._lib.load()
        n, m, e = sP.shape[0], P.shape[0], packed.shape[0]
        outF = fout if combin else fin
        # spatial_conv forward / backward: the C-ABI entries the product op surface (MCConvModule._SpatialConv) selects for
        # this layer shape, called directly so that the HIP-event time is the kernels' own (through the Python op the
        # same calls carry ~20 us of host work each): the row-per-lane entries over the list's plans for depth-wise
        # layers, mccnn_spatial_conv_fwd / _bwd (edge-streaming, or factored for Fin = 1) otherwise. What is built once
        # per neighbour list -- transposed list, row plans -- is built before the timed calls, as every further layer
        # over the same list sees it.
        conv_args = (ptr(sP), ptr(sF), ptr(sB), ptr(pdfs), ptr(P), ptr(start), ptr(packed), ptr(mn), ptr(mx), ptr(w1),
                     ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3))
        gws = [torch.empty_like(t) for t in (w1, b1, w2, b2, w3, b3)]
        rows_fwd = M._rows_shape(combin, fin, sF, m, e)
        rows_bwd = M._rows_shape(combin, fin, sF, n, e, backward=True)
        start_t = perm_t = None
        t_tr = None
        if not combin:  # the transposed neighbour list of depth-wise layers: built once per neighbour list, timed on its own
            start_t = torch.empty(n + 1, dtype=torch.int32, device=device)
            perm_t = torch.empty(max(e, 1), dtype=torch.int32, device=device)
            tws = torch.empty(max(256, lib.mccnn_transpose_neighbors_workspace_bytes(n, e)), dtype=torch.uint8, device=device)
            t_tr, _ = ev_time(lambda: check(lib.mccnn_transpose_neighbors(ptr(packed), e, n, ptr(start_t), ptr(perm_t),
                                                                           ptr(tws), tws.numel(), stream_handle()),
                                            "transpose_neighbors"))

        def plan_of(transposed):
            t0 = time.perf_counter()
            pl = M._row_plan(packed, transposed, sP, sB, pdfs, P, start, packed, mn, mx, n, m, e, B, r, False, True,
                             centre_points=P)
            torch.cuda.synchronize()
            return pl, (time.perf_counter() - t0) * 1e3

        def conv_times(feats_t, og_t, bf):
            """(fwd ms, bwd ms, feature gradient) of this layer with rows stored as f32 (bf = 0) or bf16 (bf = 1)."""
            o = torch.empty((m, outF), dtype=feats_t.dtype, device=device)
            fgr = torch.empty_like(feats_t)
            if rows_fwd:
                pf, _ = plan_of(False)
                scr = torch.empty((pf.scratch_rows, outF), dtype=torch.float32, device=device)
                tf, _ = ev_time(lambda: check(lib.mccnn_spatial_conv_fwd_rows(
                    conv_args[0], ptr(feats_t), *conv_args[2:], n, m, e, fin, B, r, 0, 1, bf, pf.vrow, pf.vcode,
                    pf.slice_off, pf.vpos_row, pf.rec, pf.other, ptr(o), ptr(scr), None, stream_handle()),
                    "conv_fwd_rows"))
                state = None
            else:
                fwd_ws = torch.empty(max(256, lib.mccnn_spatial_conv_fwd_workspace_bytes(m, e, fin, fout, int(combin))),
                                     dtype=torch.uint8, device=device)
                sbytes = lib.mccnn_spatial_conv_state_bytes(m, e, fin, fout, int(combin)) if not bf else 0
                state = torch.empty(sbytes, dtype=torch.uint8, device=device) if sbytes else None  # kept fwd -> bwd, as autograd does
                if bf:
                    tf, _ = ev_time(lambda: check(lib.mccnn_spatial_conv_fwd_bf16(
                        conv_args[0], ptr(feats_t), *conv_args[2:], n, m, e, fin, B, r, 0, 1, ptr(o), ptr(fwd_ws), fwd_ws.numel(),
                        stream_handle()), "conv_fwd_bf16"))
                else:
                    tf, _ = ev_time(lambda: check(lib.mccnn_spatial_conv_fwd(
                        *conv_args, n, m, e, fin, fout, int(combin), B, r, 0, 1, ptr(o), ptr(state), ptr(fwd_ws), fwd_ws.numel(),
                        stream_handle()), "conv_fwd"))
            if rows_bwd:
                pt, _ = plan_of(True)
                scr = torch.empty((pt.scratch_rows, fin), dtype=torch.float32, device=device)
                bws = torch.empty(max(256, lib.mccnn_spatial_conv_bwd_rows_workspace_bytes(n, e, fin)), dtype=torch.uint8, device=device)
                tb, _ = ev_time(lambda: check(lib.mccnn_spatial_conv_bwd_rows(
                    conv_args[0], ptr(feats_t), *conv_args[2:], ptr(og_t), n, m, e, fin, B, r, 0, 1, bf, ptr(pt.row_start),
                    pt.vrow, pt.vcode, pt.slice_off, pt.vpos_row, pt.rec, pt.other, ptr(fgr),
                    ptr(scr), *[ptr(g) for g in gws], None, ptr(bws), bws.numel(), stream_handle()), "conv_bwd_rows"))
            else:
                bwd_ws = torch.empty(max(256, lib.mccnn_spatial_conv_bwd_workspace_bytes(n, m, e, fin, fout, int(combin))),
                                     dtype=torch.uint8, device=device)
                if bf:
                    tb, _ = ev_time(lambda: check(lib.mccnn_spatial_conv_bwd_bf16(
                        conv_args[0], ptr(feats_t), *conv_args[2:], ptr(og_t), n, m, e, fin, B, r, 0, 1, ptr(start_t), ptr(perm_t),
                        ptr(fgr), *[ptr(g) for g in gws], ptr(bwd_ws), bwd_ws.numel(), stream_handle()), "conv_bwd_bf16"))
                else:
                    tb, _ = ev_time(lambda: check(lib.mccnn_spatial_conv_bwd(
                        *conv_args, ptr(og_t), n, m, e, fin, fout, int(combin), B, r, 0, 1, ptr(state), ptr(start_t), ptr(perm_t),
                        ptr(fgr), *[ptr(g) for g in gws], ptr(bwd_ws), bwd_ws.numel(), stream_handle()), "conv_bwd"))
            return tf, tb, fgr
        t_fwd, t_bwd, fg = conv_times(sF, self.OG, 0)
        plan_ms = None
        if rows_fwd or rows_bwd:  # one-off cost per neighbour list (host + device, synchronised): rebuilt from scratch
            packed._mccnn_rowplans = {}
            plan_ms = {}
            if rows_fwd:
                plan_ms["forward_plan_ms"] = round(plan_of(False)[1], 4)
            if rows_bwd:
                plan_ms["transposed_plan_ms_excl_transposed_list"] = round(plan_of(True)[1], 4)
        # combin layers with 2..4 input features: with the neighbour list's transposed form at hand (a training loop gets it
        # from ConvolutionBuilder.prefetch_geometry for free, on the side stream) the per-edge feature gradients are
        # gathered in a fixed order instead of added with E x Fin float atomics -- deterministic; timed beside the default
        det = None
        if combin and 2 <= fin <= 4:
            start_t = torch.empty(n + 1, dtype=torch.int32, device=device)
            perm_t = torch.empty(max(e, 1), dtype=torch.int32, device=device)
            tws = torch.empty(max(256, lib.mccnn_transpose_neighbors_workspace_bytes(n, e)), dtype=torch.uint8, device=device)
            t_tr2, _ = ev_time(lambda: check(lib.mccnn_transpose_neighbors(ptr(packed), e, n, ptr(start_t), ptr(perm_t),
                                                                            ptr(tws), tws.numel(), stream_handle()),
                                             "transpose_neighbors"))
            _, t_bdet, _ = conv_times(sF, self.OG, 0)
            det = {"bwd_ms": round(t_bdet, 4), "transposed_list_ms_once_per_neighbour_list": round(t_tr2, 4),
                   "note": "feature gradient gathered through the transposed list: no float atomics, bit-reproducible"}
            start_t = perm_t = None
        # depth-wise layers with bf16 feature rows (extension, BASELINE cfg3): same launches, rows stored as bf16
        bf16 = None
        if not combin and fin % 8 == 0:
            t_f16, t_b16, _ = conv_times(sF.to(torch.bfloat16), self.OG.to(torch.bfloat16), 1)
            bf16 = {"fwd_ms": round(t_f16, 4), "bwd_ms": round(t_b16, 4),
                    "note": "features, outputs and their gradients stored as bf16 rows; MLP and accumulation f32"}
        kernels = ("row-per-lane (conv_rows.hip)" if (rows_fwd or rows_bwd) else
                   ("factored Fin = 1 (conv_f1.hip)" if (combin and fin == 1) else "edge-streaming (conv.hip)"))
        t_s2g, _ = ev_time(lambda: M._gather_rows(fg, idx, n))
        C = int(np.prod(cells.shape[:4]))
        # algorithmic work per launch (stated in NOTES.md section 6)
        alg = {
            "sort_points_step1": ("hbm", n * 16 + n * 8 + 4 * C, t_s1),
            "sort_points_step2": ("hbm", n * (16 + 8 + 4 * fin) + n * (16 + 4 * fin) + 8 * C, t_s2),
            "find_neighbors": ("hbm", 16 * m + 12 * n + 8 * C + 4 * m + 8 * e, t_fn),
            "compute_pdf": ("hbm", 24 * e, t_pdf),
            "spatial_conv_fwd": conv_work("fwd", fin, fout, combin, nb, n, m, e) + (t_fwd,),
            "spatial_conv_bwd": conv_work("bwd", fin, fout, combin, nb, n, m, e) + (t_bwd,),
            "sort_points_step2_grad": ("hbm", n * (4 + 8 * fin), t_s2g),
        }
        if t_tr is not None:
            alg["transpose_neighbors"] = ("hbm", 8 * e + 4 * e + 4 * n, t_tr)
        breakdown = {}
        if bf16 is not None:
            breakdown["spatial_conv_bf16_rows"] = bf16
        for k, (bound, work, ms) in alg.items():
            ach = work / (ms * 1e-3) / (1e9 if bound == "hbm" else 1e12)
            breakdown[k] = {"ms": round(ms, 4), "bound": bound, "achieved": round(ach, 3),
                            "unit": "GB/s" if bound == "hbm" else "TFLOP/s"}
            if bound == "hbm" and k.startswith("spatial_conv_"):
                # wide depth-wise layers are priced by their gather bytes; the kernel-MLP rate of the same
                # launch is printed beside it -- at sizes whose rows stay in the Infinity Cache it is the tighter bound
                flops = (320.0 if k.endswith("fwd") else 912.0) * nb * e
                breakdown[k]["mlp_tflops"] = round(flops / (ms * 1e-3) / 1e12, 3)
                breakdown[k]["mlp_frac_of_f32_peak"] = round(flops / (ms * 1e-3) / 1e12 / F32_PEAK_TFLOPS, 4)
        # compute_pdf is VALU / transcendental-bound, not byte-bound: its natural rate is pair terms per second,
        # sum over the centres of k_i^2
        kk = torch.diff(torch.cat([start.reshape(-1).to(torch.int64),
                                   torch.tensor([e], dtype=torch.int64, device=device)]))
        pairs = float((kk * kk).sum().item())
        breakdown["compute_pdf"]["pair_terms"] = int(pairs)
        breakdown["compute_pdf"]["gpairs_per_s"] = round(pairs / (t_pdf * 1e-3) / 1e9, 1)
        breakdown["compute_pdf"]["bound"] = "valu+mfma issue (one v_exp_f32 per pair, one 16x16x4 MFMA per 256 pairs); GB/s of the 24 E algorithmic bytes for reference"
        dom = max(alg, key=lambda k: alg[k][2])
        bound, work, ms = alg[dom]
        peak = HBM_PEAK_GBS if bound == "hbm" else F32_PEAK_TFLOPS
        ach = work / (ms * 1e-3) / (1e9 if bound == "hbm" else 1e12)
        roofline = {"kernel": dom, "bound": bound, "achieved": round(ach, 3), "peak": peak,
                    "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                    "ms": round(ms, 4), "edges": e, "mlp_blocks": nb, "conv_kernels": kernels}
        if plan_ms:
            roofline["row_plans_once_per_neighbour_list"] = plan_ms
        if det:
            roofline["deterministic_feature_gradient"] = det
        if bound == "hbm" and dom.startswith("spatial_conv_"):
            # wide depth-wise layer on ONE room: its gathered rows (SURVEY 8d prices the layer by them) are Infinity-Cache
            # hits, not HBM traffic (counter traffic is ~1/3 of the algorithmic bytes), so the HBM peak is the wrong
            # yardstick: the kernel-MLP issue rate bounds the launch. `frac` is the MLP fraction of the f32 peak; the
            # gather rate of the algorithmic bytes is kept beside it.
            flops = (320.0 if dom.endswith("fwd") else 912.0) * nb * e
            roofline.update({"bound": "cache-gather", "achieved": round(flops / (ms * 1e-3) / 1e12, 3), "peak": F32_PEAK_TFLOPS,
                             "unit": "TFLOP/s (kernel MLP)", "frac": round(flops / (ms * 1e-3) / 1e12 / F32_PEAK_TFLOPS, 4),
                             "gather_GBs_of_algorithmic_bytes": round(ach, 1),
                             "gather_frac_of_hbm_peak": round(ach / HBM_PEAK_GBS, 4)})
        if combin and fin == 1:
            # one-input-feature layers run the factored kernels (conv_f1.hip): layer 3 is applied per centre, so fewer
            # flops are EXECUTED than the algorithm of SURVEY 8d counts; `achieved` keeps the contract's algorithmic
            # figure (an effective rate), this is the rate of the arithmetic actually issued
            ex = {"spatial_conv_fwd": 224.0 * nb * e + 144.0 * nb * m, "spatial_conv_bwd": 560.0 * nb * e + 290.0 * nb * m}
            if dom in ex:
                roofline["executed_tflops"] = round(ex[dom] / (ms * 1e-3) / 1e12, 3)
                roofline["executed_frac"] = round(roofline["executed_tflops"] / peak, 4)
            # never read the algorithmic fraction alone: both directions, algorithmic / executed / matrix-pipe busy
            both = {}
            for k, alg_f in (("spatial_conv_fwd", 320.0), ("spatial_conv_bwd", 912.0)):
                t_ms = alg[k][2]
                both[k[13:]] = {"ms": round(t_ms, 4),
                                "algorithmic_frac": round(alg_f * nb * e / (t_ms * 1e-3) / 1e12 / F32_PEAK_TFLOPS, 4),
                                "executed_frac": round(ex[k] / (t_ms * 1e-3) / 1e12 / F32_PEAK_TFLOPS, 4)}
            roofline["fwd_bwd"] = both
        # Two fields bench.py cannot measure itself (counters need rocprofv3 around the process): they are READ from the
        # committed profile of this command (tools/prof.sh -> profiles/<round>_*), are named for what they are, and carry
        # whether that profile was taken from the kernel sources of THIS tree (`kernel_sources_match`: sha1 of
        # mccnn_amd/csrc/*.hip recorded by tools/prof_summary.py next to the numbers). `traffic` -- the contract's key --
        # is the profile's number when the sources match and null otherwise.
        busy = mfma_busy_from_profile(self.layer)
        if busy is not None and a.points == 100000 and B == 1:
            roofline["mfma_pipe_busy_from_committed_profile"] = busy
        tfile = os.path.join(ROOT, "profiles", "%s_pmc_traffic_%s.json" % (PROFILE_ROUND, self.layer))
        if dom == "spatial_conv_bwd" and os.path.exists(tfile) and a.points == 100000 and B == 1:
            with open(tfile) as fh:
                tj = json.load(fh)
            match = tj.get("_kernel_sources_sha1") == kernel_sources_sha1()
            for kname, tr in tj.items():
                if kname.startswith(("conv_bwd_mfma", "f1_bwd_edges", "dw_bwd_rows")):
                    roofline["traffic_from_committed_profile"] = {
                        "bytes": int(tr["bytes"]), "source": "profiles/" + os.path.basename(tfile),
                        "kernel_sources_match": bool(match), "measured_in_this_run": False}
                    roofline["traffic"] = int(tr["bytes"]) if match else None
        # the north star's second kernel: the neighbour search against the HBM roofline (algorithmic bytes of SURVEY 8d)
        fn = breakdown.get("find_neighbors") if isinstance(breakdown, dict) else None
        if isinstance(fn, dict) and fn.get("unit") == "GB/s":
            roofline["find_neighbors"] = {"bound": "hbm", "ms": fn["ms"], "achieved": fn["achieved"], "peak": HBM_PEAK_GBS,
                                          "unit": "GB/s", "frac": round(fn["achieved"] / HBM_PEAK_GBS, 4)}
        return roofline, breakdown


class ConfigWorkload:
    """One BASELINE.json configuration (mccnn_amd.workloads.CONFIGS): a step builds the PointHierarchy of the batch and
    runs forward + backward of EVERY convolution of the model's graph through the builder -- shared grids / neighbour
    lists / PDFs come from ConvolutionBuilder's caches exactly as in the reference's graph (MCConvBuilder.py:349-391).
    The dense layers between the convolutions are not on the hot path: every convolution is fed seeded random features
    and out-gradients of the shape the graph gives it (as tests/test_gpu_configs.py does)."""

    def __init__(self, cfg, device, clouds=None, points=None):
        from mccnn_amd.MCConvBuilder import ConvolutionBuilder
        from mccnn_amd.workloads import config_points
        from mccnn_amd import _lib
        self.cfg, self.device = cfg, device
        if points is not None:
            self.pts_np, self.bids_np, self.B = points
        else:
            self.pts_np, self.bids_np, self.B = config_points(cfg, clouds)
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(device)
        self.P, self.Bi = t(self.pts_np), t(self.bids_np)
        self.F0 = torch.ones((self.P.shape[0], 1), dtype=torch.float32, device=device)  # ModelNet.py:168
        self.builder = ConvolutionBuilder(KDEWindow=0.25, relativeRadius=cfg.relative)
        self.lib = _lib.load()
        self.feats = self.ogs = None
        # pipelined steps: the hierarchy of the NEXT batch is requested as soon as the one of this batch is adopted
        # (PointHierarchy.prefetch: own stream, helper thread) and runs under this batch's convolutions; every step still
        # builds one hierarchy and runs every convolution
        self.pipeline = False
        self.next_ph = None
        torch.manual_seed(4321)
        self.params = None
        self.outs = self.step()  # creates variables, features and out-gradients

    def hierarchy(self, prefetched=None):
        from mccnn_amd.MCConvBuilder import PointHierarchy
        return PointHierarchy(self.P, self.F0, self.Bi, list(self.cfg.hierarchy), "PH_" + self.cfg.name, self.B,
                              self.cfg.relative, prefetched=prefetched)

    def request_next(self):
        from mccnn_amd.MCConvBuilder import PointHierarchy
        # after=True: the synthetic batch has been resident in HBM since before the timed region -- its hierarchy does not
        # queue behind the convolutions the calling stream holds (a loader would pass the event of its upload stream)
        # (features=: the input feature rows of the batch travel with it -- every level's rows are gathered on the
        # hierarchy's stream instead of by one launch per level on this thread at adoption)
        self.next_ph = PointHierarchy.prefetch(self.P, self.Bi, list(self.cfg.hierarchy), self.B, self.cfg.relative, after=True,
                                               features=(None if os.environ.get("MCCNN_BENCH_NO_FEATURE_PREFETCH") else self.F0))
        return self.next_ph is not None

    def set_pipeline(self, on, geometry=False):
        """-> whether pipelined steps are active. Switching on requests the first hierarchy ahead. geometry=True: the
        hierarchy runs TWO batches ahead, so that the one of the next batch is complete when a step starts and
        everything the step after this one needs -- grids, neighbour lists, PDFs, row plans -- is started on side streams
        under this step's convolutions (ConvolutionBuilder.prefetch_step)."""
        self.pipeline = bool(on) and self.request_next()
        self.deep = False
        self.ready_ph = None
        if not self.pipeline:
            self.next_ph = None
        elif geometry:
            self.ready_ph = self.hierarchy(self.next_ph)   # the hierarchy of the first pipelined step
            self.request_next()
            self.deep = True
        return self.pipeline

    def _make_rows(self, ph):
        self.feats_np, self.ogs_np, self.feats, self.ogs = [], [], [], []
        for ci, c in enumerate(self.cfg.convs):
            rng = np.random.default_rng(100 + ci)
            n, m = int(ph.points_[c.lin].shape[0]), int(ph.points_[c.lout].shape[0])
            outF = c.fout if c.combin else c.fin
            f = (2 * rng.random((n, c.fin)) - 1).astype(np.float32)
            g = (2 * rng.random((m, outF)) - 1).astype(np.float32)
            ft, gt = torch.from_numpy(f).to(self.device), torch.from_numpy(g).to(self.device)
            if c.bf16:
                ft, gt = ft.to(torch.bfloat16), gt.to(torch.bfloat16)
                f, g = ft.float().cpu().numpy(), gt.float().cpu().numpy()  # the CPU leg gets the same rounded values
            self.feats_np.append(f)
            self.ogs_np.append(g)
            self.feats.append(ft.requires_grad_(True))
            self.ogs.append(gt)

    def conv(self, ph, ci):
        c = self.cfg.convs[ci]
        return self.builder.create_convolution(c.name, ph, c.lin, self.feats[ci], c.fin, c.radius, ph, c.lout, c.combin,
                                               c.fout, c.window)

    def step(self):
        self.builder.reset()
        if self.pipeline and getattr(self, "deep", False):
            ph = self.ph = self.ready_ph
            nxt = self.hierarchy(self.next_ph)      # the next batch's hierarchy, requested a step ago: complete by now
            self.request_next()                     # ... and the one after it
            self.builder.prefetch_step(nxt)         # the next batch's geometry and row plans, under this batch's layers
            self.ready_ph = nxt
        elif self.pipeline:
            ph = self.ph = self.hierarchy(self.next_ph)
            self.request_next()
        else:
            ph = self.ph = self.hierarchy()
        if self.feats is None:
            self._make_rows(ph)
        outs = [self.conv(ph, ci) for ci in range(len(self.cfg.convs))]
        if getattr(self, "params", None) is None:
            self.params = list(self.builder.parameters())   # (created by the first step's layers)
        inputs = self.feats + self.params
        # gradients of every convolution w.r.t. its features and its six kernel-MLP tensors, nothing accumulated
        self.grads = torch.autograd.grad(outs, inputs, self.ogs, allow_unused=True)
        return outs

    def timed(self, steps, warmup):
        for _ in range(warmup):
            self.step()
        torch.cuda.synchronize()
        from mccnn_amd import MCConvModule as _M
        l0 = self.lib.mccnn_debug_launch_count()
        w0 = _M.host_wait_seconds()
        lw0 = _M.HOST_LAG_WAIT_S[0]
        # the host issues a step while the device is at most `lag` steps behind (0 = unbounded; run_config(): 2 for the
        # pipelined modes)
        lag = getattr(self, "lag", 0)
        self.builder.hostStepsAhead_ = (lag - 1) if lag else None   # (ConvolutionBuilder.reset() does the waiting)
        self.builder.__dict__.pop("stepEvents_", None)
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        t_issue = time.perf_counter() - t0   # the host has enqueued the last launch (edge-count waits included)
        waits = _M.host_wait_seconds() - w0
        lag_waits = _M.HOST_LAG_WAIT_S[0] - lw0
        self.builder.hostStepsAhead_ = None
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        launches = (self.lib.mccnn_debug_launch_count() - l0) / float(steps)
        self.host_issue_ms = t_issue / steps * 1e3
        # ... of which the host WAITED for sizes computed on the device (edge totals, sample counts): the rest is its own work
        self.host_wait_ms = waits / steps * 1e3
        # ... split: waiting for the DEVICE to catch up (the host is held at most `lag` steps ahead: the step is bound by
        # the GPU when this is large) and waiting for device-side SIZES (edge totals, level sizes)
        self.host_lag_wait_ms = lag_waits / steps * 1e3
        return el / steps * 1e3, launches

    def per_layer(self, iters=5):
        """HIP-event times of the hierarchy build and of every convolution's forward / backward with the geometry
        cached (what the layer itself costs; the first use of a grid / neighbour list / PDF is in the step time)."""
        def ev():
            return torch.cuda.Event(enable_timing=True)
        self.builder.reset()
        t_h = []
        for _ in range(iters):
            torch.cuda.synchronize()
            a, b = ev(), ev()
            a.record()
            ph = self.hierarchy()
            b.record()
            b.synchronize()
            t_h.append(a.elapsed_time(b))
        for ci in range(len(self.cfg.convs)):  # populate the caches
            self.conv(ph, ci)
        layers = []
        for ci, c in enumerate(self.cfg.convs):
            tf, tb = [], []
            named = dict(self.builder.named_parameters())
            inputs = [self.feats[ci]] + [named[c.name + sfx] for sfx in ("_weights", "_biases", "_weights2", "_biases2",
                                                                         "_weights3", "_biases3")]
            for it in range(iters + 1):
                torch.cuda.synchronize()
                a, b, d = ev(), ev(), ev()
                a.record()
                out = self.conv(ph, ci)
                b.record()
                torch.autograd.grad([out], inputs, [self.ogs[ci]])
                d.record()
                d.synchronize()
                if it:
                    tf.append(a.elapsed_time(b))
                    tb.append(b.elapsed_time(d))
            keyG, keyN, _ = self.builder.__compute_dic_keys__(ph, ph, c.lin, c.lout, c.radius, c.window, self.cfg.relative, True)
            e = int(self.builder.cacheNeighs_[keyN][1].shape[0])
            n, m = int(ph.points_[c.lin].shape[0]), int(ph.points_[c.lout].shape[0])
            nb = ((c.fin * c.fout if c.combin else c.fin) + 7) // 8
            ent = {"name": c.name, "levels": [c.lin, c.lout], "radius": round(c.radius, 4), "fin": c.fin, "fout": c.fout,
                   "combin": c.combin, "bf16_rows": c.bf16, "points_in": n, "centres": m, "edges": e, "mlp_blocks": nb,
                   "fwd_ms": round(float(np.mean(tf)), 4), "bwd_ms": round(float(np.mean(tb)), 4)}
            for which, ms in (("fwd", ent["fwd_ms"]), ("bwd", ent["bwd_ms"])):
                flops = (320.0 if which == "fwd" else 912.0) * nb * e
                rl = {"bound": "mfma", "achieved": round(flops / (ms * 1e-3) / 1e12, 3), "peak": F32_PEAK_TFLOPS,
                      "unit": "TFLOP/s", "frac": round(flops / (ms * 1e-3) / 1e12 / F32_PEAK_TFLOPS, 4)}
                if conv_bound(c.fin, c.combin) == "hbm":
                    _, byt = conv_work(which, c.fin, c.fout, c.combin, nb, n, m, e)
                    if c.bf16:
                        byt = byt - (e * 2 * c.fin if which == "fwd" else e * 6 * c.fin)  # half-width gathered rows
                    rl["bound"] = "cache-gather"
                    rl["gather_GBs_of_algorithmic_bytes"] = round(byt / (ms * 1e-3) / 1e9, 1)
                ent["roofline_" + which] = rl
            layers.append(ent)
        return float(np.mean(t_h)), layers, [int(p.shape[0]) for p in ph.points_]


def cpu_config(cw, sample_clouds, budget_s=8.0):
    """The CPU port (oracle, OpenMP over centres) on a BOUNDED sample of the configuration -- the first `sample_clouds`
    clouds (rooms: an x-slab with 1/16 of the points, same density) -- running the same graph: hierarchy op by op,
    then every convolution forward + backward with grids / neighbour lists / PDFs shared like the builder's caches.
    The GPU path runs the same sample once and every convolution output is compared (checker use of the oracle)."""
    from oracle.oracle import Oracle
    from mccnn_amd.workloads import config_points
    cfg = cw.cfg
    if cfg.cloud_kind == "room":
        first = cw.pts_np[:cfg.points]
        order = np.argsort(first[:, 0], kind="stable")
        k = max(len(first) // 16, 1)
        lo = (len(first) - k) // 2
        sel = np.sort(order[lo:lo + k])
        pts, bids, B = first[sel], np.zeros((k, 1), np.int32), 1
        what = "x-slab of the first room holding 1/16 of its points (%d), same density" % k
    else:
        pts, bids, B = config_points(cfg, min(sample_clouds, cfg.clouds))
        what = "the first %d of %d clouds (%d points)" % (B, cfg.clouds, len(pts))
    gw = ConfigWorkload(cfg, cw.device, points=(pts, bids, B))
    sd = {k_: v.detach().clone() for k_, v in cw.builder.state_dict().items()}
    gw.builder.load_state_dict(sd)   # the same kernel-MLP variables as the timed workload
    g_outs = [o.detach().float().cpu().numpy() for o in gw.step()]
    torch.cuda.synchronize()
    orc = Oracle(omp=True)
    W = {k_: v.detach().cpu().numpy() for k_, v in sd.items()}
    rel, f0 = cfg.relative, np.ones((len(pts), 1), np.float32)

    def cpu_step():
        mn, mx = orc.compute_aabb(pts, bids, B, rel)
        lev = [(pts, bids)]
        cp, cb, cf = pts, bids, f0
        for r in cfg.hierarchy:      # MCConvBuilder.py:101-128
            k_, i_ = orc.sort_points_step1(cp, cb, mn, mx, B, r, rel)
            sp, sb, sf, cl = orc.sort_points_step2(cp, cb, cf, k_, i_, mn, mx, B, r, rel)
            op, ob, oi = orc.poisson_sampling(sp, sb, cl, mn, mx, r, B, rel)
            cf = orc.get_sampled_features(oi, sf)
            orc.transform_indexs(oi, i_)
            cp, cb = op, ob
            lev.append((cp, cb))
        grids, neighs, pdfs, outs = {}, {}, {}, []
        for ci, c in enumerate(cfg.convs):     # MCConvBuilder.py:349-427
            kg, kn = (c.lin, c.radius), (c.lin, c.radius, c.lout)
            kp = kn + (c.window,)
            ip, ib = lev[c.lin]
            o_p, o_b = lev[c.lout]
            if kg not in grids:
                k_, i_ = orc.sort_points_step1(ip, ib, mn, mx, B, c.radius, rel)
                sp, sb, sf, cl = orc.sort_points_step2(ip, ib, cw_feats[ci], k_, i_, mn, mx, B, c.radius, rel)
                grids[kg] = (sp, sb, cl, i_)
            else:
                sf = orc.sort_features(cw_feats[ci], grids[kg][3])
            sp, sb, cl, i_ = grids[kg]
            if kn not in neighs:
                neighs[kn] = orc.find_neighbors(o_p, o_b, sp, cl, mn, mx, c.radius, B, rel)
            st, pk = neighs[kn]
            if kp not in pdfs:
                pdfs[kp] = orc.compute_pdf(sp, sb, mn, mx, st, pk, c.window, c.radius, B, rel)
            nm = c.name
            a_ = (sp, sf, sb, pdfs[kp], o_p, st, pk, mn, mx, W[nm + "_weights"], W[nm + "_weights2"].reshape(8, -1),
                  W[nm + "_weights3"].reshape(8, -1), W[nm + "_biases"], W[nm + "_biases2"].reshape(-1),
                  W[nm + "_biases3"].reshape(-1))
            outs.append(orc.spatial_conv(*a_, c.fout, c.combin, B, c.radius, rel, True))
            g_ = orc.spatial_conv_grad(*a_, cw_ogs[ci], c.fout, c.combin, B, c.radius, rel, True)
            orc.sort_points_step2_grad(i_, np.zeros_like(sp), g_[0])
        return outs

    cw_feats, cw_ogs = gw.feats_np, gw.ogs_np
    outs = cpu_step()  # warm-up (and the outputs the GPU sample is checked against)
    ts = []
    t0 = time.perf_counter()
    while len(ts) < 3 and (not ts or time.perf_counter() - t0 < budget_s):
        c0 = time.perf_counter()
        cpu_step()
        ts.append(time.perf_counter() - c0)
    med = float(np.median(ts))
    worst = 0.0
    for ci, (go, co) in enumerate(zip(g_outs, outs)):
        tol_scale = max(float(np.abs(co).max()), 1e-30)
        worst = max(worst, float(np.abs(go - co).max() / tol_scale) if not cfg.convs[ci].bf16 else 0.0)
    return {"value": round(len(pts) / med, 1), "unit": "points/s", "cores": orc.num_threads(), "kind": "port",
            "sample": "%s; median of %d steps after 1 warm-up, %.2f s per step" % (what, len(ts), med),
            "gpu_vs_oracle_max_rel_err_f32_layers": float("%.3e" % worst)}


def run_config(name, device, args, want_cpu):
    from mccnn_amd.workloads import CONFIGS
    cfg = CONFIGS[name]
    cw = ConfigWorkload(cfg, device)
    steps = {"cfg0": 200, "cfg1": 100, "cfg2": 40, "cfg3": 30, "cfg4": 30}[name]
    ms_seq, launches = cw.timed(steps, 5)
    seq_issue, seq_wait = cw.host_issue_ms, cw.host_wait_ms
    ms, mode, lag_used = ms_seq, "sequential", 0
    if not getattr(args, "no_pipeline", False) and cfg.hierarchy:
        # the same steps with the hierarchy of the next batch requested one step ahead -- accepted when the outputs of
        # every convolution are bit-identical to the sequential step's and the steps are not slower
        try:
            ref = [o.detach().clone() for o in cw.step()]
            best_issue, best_wait, best_lag = seq_issue, seq_wait, 0.0
            # ONE pipelined form is timed: hierarchy two batches ahead + the next batch's geometry (MCCNN_BENCH_DEEP=0: the
            # shallow form, hierarchy one batch ahead, instead)
            for deep in ((True,) if os.environ.get("MCCNN_BENCH_DEEP", "1") != "0" else (False,)):
                cw.set_pipeline(False)
                torch.cuda.synchronize()
                if not cw.set_pipeline(True, geometry=deep):
                    break
                same = True
                for _ in range(4):
                    same = same and all(torch.equal(a, b) for a, b in zip(cw.step(), ref))
                if same:
                    # the host at most ONE step ahead of the device (ConvolutionBuilder.hostStepsAhead_ = 1), as a training
                    # loop would run: bounded memory, and no slower than letting it run free (r05, tools/lag_probe.py:
                    # cfg2 1.43 against 1.49 ms unbounded and 1.67 one step at a time, cfg3 5.90 / 6.01 / 6.10)
                    lag = int(os.environ.get("MCCNN_BENCH_LAG", "2"))
                    cw.lag = lag
                    for _ in range(3):
                        cw.step()
                    ms_p, launches_p = cw.timed(steps, 5)
                    cw.lag = 0
                    if os.environ.get("MCCNN_BENCH_VERBOSE"):
                        print("bench: %s %s host lag %d: %.4f ms" % (name, "deep" if deep else "pipelined", lag, ms_p), file=sys.stderr)
                    if ms_p < ms:
                        ms, launches, mode, lag_used = ms_p, launches_p, ("pipelined+geometry" if deep else "pipelined"), lag
                        best_issue, best_wait, best_lag = cw.host_issue_ms, cw.host_wait_ms, cw.host_lag_wait_ms
                else:
                    print("bench: %s steps of %s do not reproduce the sequential outputs" % (
                        "pipelined+geometry" if deep else "pipelined", name), file=sys.stderr)
            cw.host_issue_ms, cw.host_wait_ms, cw.host_lag_wait_ms = best_issue, best_wait, best_lag
        except Exception as ex:  # the sequential numbers stand
            print("bench: pipelined %s steps failed: %r" % (name, ex), file=sys.stderr)
            cw.host_issue_ms, cw.host_wait_ms, cw.host_lag_wait_ms = seq_issue, seq_wait, 0.0
        cw.set_pipeline(False)
        torch.cuda.synchronize()
    n = int(cw.P.shape[0])
    t_h, layers, sizes = cw.per_layer()
    ent = {"workload": cfg.what, "points": n, "clouds": cw.B, "level_sizes": sizes, "convolutions": len(cfg.convs),
           "steps": steps, "ms_per_step": round(ms, 4), "value": round(n / (ms * 1e-3), 1), "unit": "points/s",
           # "pipelined": the next batch's PointHierarchy is requested one step ahead (PointHierarchy.prefetch) and built
           # under this batch's convolutions; "pipelined+geometry": two batches ahead, and the next batch's grids / lists /
           # PDFs / row plans are started under this batch's convolutions as well (ConvolutionBuilder.prefetch_step);
           # sequential_ms_per_step: hierarchy, then convolutions, nothing carried over
           "mode": mode, "sequential_ms_per_step": round(ms_seq, 4),
           # pipelined modes: the host issues a step while the device is at most this many steps behind
           # (ConvolutionBuilder.hostStepsAhead_ + 1); 0 = sequential steps, nothing bounded
           "host_lag_steps": lag_used,
           "library_launches_per_step": round(launches, 1),
           # when this equals ms_per_step the step is bound by the HOST issuing its launches, not by the kernels
           "host_issue_ms_per_step": round(cw.host_issue_ms, 4),
           # the host's OWN work per step: issue time minus the time it sat waiting for device-side sizes
           "host_busy_ms_per_step": round(cw.host_issue_ms - cw.host_wait_ms, 4),
           # the waits, split: held back by the lag bound (= the GPU is the bound) / waiting for device-side sizes
           "host_lag_wait_ms_per_step": round(getattr(cw, "host_lag_wait_ms", 0.0), 4),
           "host_size_wait_ms_per_step": round(cw.host_wait_ms - getattr(cw, "host_lag_wait_ms", 0.0), 4),
           # who bounds the step: the host's own work against what is left of the step
           "bound_by": ("host" if (cw.host_issue_ms - cw.host_wait_ms) >= 0.85 * ms else "gpu"),
           # the prefetched hierarchy of the pipelined modes starts at once (after=True: the synthetic batch is resident in
           # HBM before the timed region); a loader that uploads per step would pass the event of its upload stream
           "hierarchy_start": ("after=True (batch resident in HBM)" if mode != "sequential" else "inline"),
           "hierarchy_ms": round(t_h, 4),
           "conv_fwd_bwd_ms_cached_geometry": round(sum(l["fwd_ms"] + l["bwd_ms"] for l in layers), 4),
           "layers": layers}
    if want_cpu == "later":   # main(): every configuration's GPU steps first, the 128-thread CPU legs after all of them
        return ent, cw
    if want_cpu:
        config_cpu_leg(ent, cw, name)
    del cw
    torch.cuda.empty_cache()
    return ent


def config_cpu_leg(ent, cw, name):
    try:
        ent["cpu_baseline"] = cpu_config(cw, {"cfg0": 1, "cfg1": 8, "cfg2": 4, "cfg3": 2, "cfg4": 1}[name])
    except Exception as ex:  # informative; never fail the bench on it
        ent["cpu_baseline"] = {"error": repr(ex)}


def kernel_sources_sha1():
    """sha1 over mccnn_amd/csrc/*.hip and *.h (sorted by name): recorded by tools/prof_summary.py in every committed
    profile, compared here so that a number read from a profile says whether it belongs to the kernels of this tree."""
    import glob
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "mccnn_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def mfma_busy_from_profile(layer):
    """Matrix-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)) of the main forward /
    backward kernels of `layer`, from the committed rocprofv3 --pmc pass of this command (tools/prof.sh); bench.py
    cannot collect counters itself. None when the profile is absent."""
    path = os.path.join(ROOT, "profiles", "%s_kernels_%s.json" % (PROFILE_ROUND, layer))
    if not os.path.exists(path):
        return None
    try:
        with open(path) as fh:
            ks = json.load(fh)
    except (OSError, ValueError):
        return None
    out = {}
    for name, rec in ks.items():
        if not isinstance(rec, dict) or "mfma_busy" not in rec:
            continue
        if name.startswith(("f1_fwd_edges", "conv_stream", "dw_fwd")) and "fwd" not in out:
            out["fwd"] = {"kernel": name.split("(")[0][:60], "busy": rec["mfma_busy"]}
        if name.startswith(("f1_bwd_edges", "conv_bwd_mfma", "dw_bwd")) and "bwd" not in out:
            out["bwd"] = {"kernel": name.split("(")[0][:60], "busy": rec["mfma_busy"]}
    if not out:
        return None
    out["source"] = "profiles/" + os.path.basename(path)
    out["kernel_sources_match"] = bool(ks.get("_kernel_sources_sha1") == kernel_sources_sha1())
    out["measured_in_this_run"] = False
    return out


def cpu_baseline(wl, out_gpu):
    """the oracle's OpenMP build on all host cores, identical workload, median of 5 steps after one warm-up;
    plus a single-thread figure on a bounded sample (a slab of the same room holding 1/16 of the points)."""
    from oracle.oracle import Oracle
    a = wl.args
    ws = [p.detach().cpu().numpy() for p in wl.builder.parameters()]
    w1, b1, w2, b2, w3, b3 = ws[0], ws[1], ws[2].reshape(8, -1), ws[3].reshape(-1), ws[4].reshape(8, -1), ws[5].reshape(-1)

    def make_step(orc, pts, bids, feats, ograd, B):
        mn, mx = orc.compute_aabb(pts, bids, B, False)

        def cpu_step():
            k, i = orc.sort_points_step1(pts, bids, mn, mx, B, a.radius, False)
            sp, sb, sf, cl = orc.sort_points_step2(pts, bids, feats, k, i, mn, mx, B, a.radius, False)
            st, pk = orc.find_neighbors(pts, bids, sp, cl, mn, mx, a.radius, B, False)
            pdf = orc.compute_pdf(sp, sb, mn, mx, st, pk, a.window, a.radius, B, False)
            arg = (sp, sf, sb, pdf, pts, st, pk, mn, mx, w1, w2, w3, b1, b2, b3)
            oc = orc.spatial_conv(*arg, wl.fout, wl.combin, B, a.radius, False, True)
            g = orc.spatial_conv_grad(*arg, ograd, wl.fout, wl.combin, B, a.radius, False, True)
            orc.sort_points_step2_grad(i, np.zeros_like(sp), g[0])
            return oc
        return cpu_step

    def median_of(fn, runs, budget_s):
        oc = fn()  # warm-up
        ts = []
        t0 = time.perf_counter()
        while len(ts) < runs and (len(ts) < 1 or time.perf_counter() - t0 < budget_s):
            c0 = time.perf_counter()
            oc = fn()
            ts.append(time.perf_counter() - c0)
        return float(np.median(ts)), len(ts), oc

    orc = Oracle(omp=True)
    step = make_step(orc, wl.pts_np, wl.bids_np, wl.feats_np, wl.ograd_np, wl.B)
    med, runs, oc = median_of(step, 5, 25.0)
    m_local = len(wl.pts_np)
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), "")
    except OSError:
        pass
    cpu = {"value": round(m_local / med, 1), "unit": "points/s", "cores": orc.num_threads(), "kind": "port",
           "host": "%s, %d logical cores" % (cpu_model, os.cpu_count() or 0),
           "sample": "median of %d steps (fwd+bwd) of the identical workload after 1 warm-up, OpenMP over centres, "
                     "%.2f s per step" % (runs, med)}
    err = float(np.abs(out_gpu.detach().cpu().numpy() - oc).max() / max(np.abs(oc).max(), 1e-30))
    cpu["gpu_vs_oracle_max_rel_err"] = float("%.3e" % err)
    # single thread: a slab of the first room along x holding 1/16 of its points (same density, so the same number of
    # neighbours per point away from the two cut faces)
    try:
        first = wl.pts_np[:a.points]
        order = np.argsort(first[:, 0], kind="stable")
        k = max(len(first) // 16, 1)
        lo = (len(first) - k) // 2
        sel = np.sort(order[lo:lo + k])
        orc1 = Oracle(omp=False)
        step1 = make_step(orc1, first[sel], np.zeros((k, 1), np.int32), wl.feats_np[:a.points][sel],
                          wl.ograd_np[:a.points][sel], 1)
        med1, runs1, _ = median_of(step1, 3, 20.0)
        cpu["single_thread"] = {"value": round(k / med1, 1), "unit": "points/s", "cores": 1,
                                "sample": "median of %d steps on a %d-point x-slab of the room (1/16 of the points, "
                                          "same density), %.2f s per step" % (runs1, k, med1)}
    except Exception as ex:  # informative only
        cpu["single_thread"] = {"error": repr(ex)}
    return cpu


LINE_LIMIT = 4096  # bytes of the final stdout line: the driver's parser gave up on the 32 KB line of round 3


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_record(rec, details_path=None):
    """The ONE line the driver parses: headline, roofline, cpu_baseline, strong and a five-object summary of the
    BASELINE configurations. Per-layer lists, the per-op breakdown and every explanatory string live in the details
    record (bench_details.json and an EARLIER stdout line)."""
    cfg = rec.get("config") or {}
    out = {k: rec.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                   "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    if isinstance(rec.get("sequential"), dict):   # the strictly sequential step (SURVEY 8d's literal definition) beside `value`
        out["sequential"] = _pick(rec["sequential"], ("ms_per_step", "value", "unit"))
    out["config"] = _pick(cfg, ("workload", "points_total", "points_per_gpu", "edges_per_gpu", "layer", "parallelism",
                                "headline_mode", "pipelined_ms_per_step", "sequential_ms_per_step",
                                "collective_backend", "rccl_world_size", "collectives_warmed_before_warmup"))
    if isinstance(out["config"].get("workload"), str):
        out["config"]["workload"] = out["config"]["workload"][:300]
    rl = rec.get("roofline")
    if isinstance(rl, dict):
        r = _pick(rl, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "ms", "executed_frac"))
        busy = rl.get("mfma_pipe_busy_from_committed_profile") or {}
        if busy:
            r["mfma_pipe_busy_from_committed_profile"] = {d: busy[d].get("busy") for d in ("fwd", "bwd") if isinstance(busy.get(d), dict)}
            r["mfma_pipe_busy_from_committed_profile"]["kernel_sources_match"] = busy.get("kernel_sources_match")
        if isinstance(rl.get("traffic_from_committed_profile"), dict):
            r["traffic_from_committed_profile"] = _pick(rl["traffic_from_committed_profile"], ("bytes", "source", "kernel_sources_match"))
        fb = rl.get("fwd_bwd") or {}
        if fb:
            r["fwd_bwd"] = {d: _pick(fb[d], ("ms", "algorithmic_frac")) for d in ("fwd", "bwd") if d in fb}
        for k in ("find_neighbors", "find_neighbors_8rooms"):
            if isinstance(rl.get(k), dict):
                r[k] = _pick(rl[k], ("bound", "ms", "achieved", "peak", "unit", "frac", "rooms", "edges"))
        out["roofline"] = r
    else:
        out["roofline"] = rl
    cb = rec.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, ("value", "unit", "cores", "kind", "sample", "error"))
        for k in ("sample", "error"):
            if k in c:
                c[k] = str(c[k])[:200]
        if isinstance(cb.get("single_thread"), dict):
            c["single_thread"] = _pick(cb["single_thread"], ("value", "cores"))
        out["cpu_baseline"] = c
    else:
        out["cpu_baseline"] = cb
    st = rec.get("strong")
    out["strong"] = _pick(st, ("rooms", "points_total", "value", "ms_per_step", "unit", "scaling", "mode")) if st else st
    ly = rec.get("layers")
    if isinstance(ly, dict):
        out["layers"] = {}
        for name, ent in ly.items():
            e = _pick(ent, ("ms_per_step", "value", "mode", "sequential_ms_per_step"))
            if isinstance(ent.get("conv_ms"), dict):
                e["conv_ms"] = ent["conv_ms"]
            if isinstance(ent.get("roofline"), dict):
                e["frac"] = ent["roofline"].get("frac")
                e["bound"] = ent["roofline"].get("bound")
            out["layers"][name] = e
    cf = rec.get("configs")
    if isinstance(cf, dict):
        out["configs"] = {}
        for name, ent in cf.items():
            e = _pick(ent, ("ms_per_step", "value", "mode", "host_lag_steps", "sequential_ms_per_step", "host_issue_ms_per_step",
                            "host_busy_ms_per_step", "host_lag_wait_ms_per_step", "bound_by", "library_launches_per_step", "points",
                            "convolutions"))
            if "library_launches_per_step" in e:
                e["launches"] = e.pop("library_launches_per_step")
            if isinstance(ent.get("cpu_baseline"), dict) and "value" in ent["cpu_baseline"]:
                e["cpu_value"] = ent["cpu_baseline"]["value"]
            if "error" in ent:
                e["error"] = str(ent["error"])[:120]
            out["configs"][name] = e
    if details_path:
        out["details"] = details_path
    line = json.dumps(out, separators=(",", ":"))
    # never let the line grow past the limit again: drop the optional objects, least important first
    for k in ("layers", "strong", "configs"):
        if len(line) < LINE_LIMIT:
            break
        out.pop(k, None)
        line = json.dumps(out, separators=(",", ":"))
    if len(line) >= LINE_LIMIT:
        raise AssertionError("bench line of %d bytes" % len(line))
    return line


def emit_record(rec, details_path):
    """Full record -> `details_path` (and one EARLIER stdout line prefixed by `details:`), compact record -> the LAST
    stdout line."""
    full = json.dumps(rec)
    written = None
    if details_path:
        try:
            with open(details_path, "w") as f:
                f.write(full + "\n")
            written = details_path
        except OSError as ex:
            log("could not write %s: %r" % (details_path, ex))
    print("details: " + full, flush=True)
    print(compact_record(rec, written), flush=True)


def self_launch(n):
    """Re-run this command as n ranks of `python -m torch.distributed.run` on a free local port and return its exit
    code. With fewer GPUs than ranks the ranks share the devices and the collectives go through gloo (RCCL refuses two
    ranks on one device); that form only exercises the N > 1 code path, the driver's 8-GPU node gets RCCL."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.device_count() < n:
        env.setdefault("MCCNN_BENCH_BACKEND", "gloo")
        log("bench.py: %d ranks on %d GPU(s): ranks share devices, collectives over gloo" % (n, torch.cuda.device_count()))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--details", default=os.path.join(ROOT, "bench_details.json"),
                    help="file for the full record (per-layer lists, per-op breakdown); '' = stdout only")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # 0.6 ms each: a 20-step region is too short to be stable
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--points", type=int, default=100000, help="points per room")
    ap.add_argument("--rooms-per-gpu", type=int, default=1, help="weak scaling: rooms on every rank")
    ap.add_argument("--scaling", choices=("both", "weak", "strong"), default="both",
                    help="both (default): headline = weak (one room per rank), the fixed batch of --strong-rooms rooms split "
                         "over the ranks is measured beside it (`strong` object); weak / strong: only that one")
    ap.add_argument("--strong-rooms", type=int, default=8, help="strong scaling: rooms in the fixed batch")
    ap.add_argument("--layer", choices=sorted(LAYERS), default="1to64")
    ap.add_argument("--radius", type=float, default=0.1)
    ap.add_argument("--window", type=float, default=0.2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--no-layers", action="store_true", help="skip the per-layer-shape measurements")
    ap.add_argument("--config", choices=("cfg0", "cfg1", "cfg2", "cfg3", "cfg4"), default=None,
                    help="measure only this BASELINE.json configuration in the `configs` object (default: all five)")
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE.json configuration measurements")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="strictly sequential steps. Default: the geometry (grid build, search, KDE) of batch k+1 runs on a "
                         "side stream under the convolutions of batch k (ConvolutionBuilder.prefetch_geometry); every step "
                         "still executes all of it, and the sequential time is reported beside the headline")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: the reference has no launcher of its own (ModelNet/ModelNet.py:202-203 pins
        # one session to one GPU), so the bench carries one -- N ranks of this very command under torch.distributed.run
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        # a launcher IS present and disagrees: --gpus N is only valid under a launcher that started N ranks
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d -- launch N > 1 as `python -m torch.distributed.run "
                         "--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`"
                         % (args.gpus, world))
    # one process per GPU; MCCNN_BENCH_BACKEND=gloo lets several ranks share one GPU (single-GPU smoke test of the
    # N > 1 code path only -- the real runs use nccl == RCCL over xGMI)
    backend = os.environ.get("MCCNN_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # MCCNN_BENCH_FORCE_PG=1: a process group of ONE rank, so that a 1-GPU box exercises RCCL's init, the asynchronous
    # gradient all-reduce and the barriers of the N > 1 path (tests/test_gpu_dist.py)
    dist_on = world > 1 or os.environ.get("MCCNN_BENCH_FORCE_PG") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != world:
            raise SystemExit("bench.py: process group of %d ranks, WORLD_SIZE %d" % (dist.get_world_size(), world))
        # Every collective KIND the steps use runs once here, before any warm-up or timed step: RCCL builds its rings /
        # channels and loads its kernels on the first call of a (collective, dtype, op), which takes hundreds of
        # milliseconds over xGMI -- a SCALE run must time the steady state, whatever --warmup says.
        for op_ in (dist.ReduceOp.SUM, dist.ReduceOp.MIN, dist.ReduceOp.MAX):      # gradients; whole-batch box (dist.py)
            w_ = torch.zeros(4096, dtype=torch.float32, device=device)
            dist.all_reduce(w_, op=op_)
        w64 = torch.zeros(3, dtype=torch.float64, device=device)                   # the timing reductions of timed()
        dist.all_reduce(w64, op=dist.ReduceOp.MAX)
        dist.all_gather([torch.zeros_like(w64) for _ in range(world)], w64)
        dist.barrier()
        torch.cuda.synchronize()

    # one process per GPU: the autograd engine's per-device worker thread only adds a thread hand-off (~0.2 ms per
    # backward() call, more than a quarter of a step here); run the backward pass on the calling thread
    if hasattr(torch.autograd, "set_multithreading_enabled"):
        torch.autograd.set_multithreading_enabled(False)

    from mccnn_amd import build as mbuild
    if rank == 0:
        mbuild.build()   # (no-op when the library and the extension are newer than their sources)
    if dist_on:
        dist.barrier()

    from mccnn_amd.dist import cloud_partition

    def strong_seeds():
        first, last = cloud_partition(args.strong_rooms, world)[rank]
        return [20180601 + r for r in range(first, last)]

    # which rooms this rank owns: weak = its own rooms_per_gpu rooms; strong = its share of a fixed batch
    if args.scaling == "strong":
        seeds = strong_seeds()
        if not seeds:
            raise SystemExit("--scaling strong needs --strong-rooms >= number of ranks")
    else:
        seeds = [20180601 + rank * args.rooms_per_gpu + r for r in range(args.rooms_per_gpu)]

    wl = Workload(args, args.layer, seeds, rank, world, device)

    # Order of the measurements. After a second of idle (a fresh process, or the host-side set-up of another workload)
    # the GPU needs ~20 ms of sustained load before it holds its clocks again: a block of 20 steps timed right after
    # 4 warm-up steps reads 0.827 ms/step where every later block of the same process reads 0.778 (tools/steps_probe.py).
    # So all workloads are SET UP first (host work), then the step loops run back to back -- the other layer shapes,
    # then the headline region (W warm-up steps, exactly K timed steps between barriers), then the per-op breakdowns
    # (rank 0; they drain the queue around every op), and the 8-room batch of the strong-scaling point last. (The headline
    # region used to follow the 8-room loop: after its 5 ms steps the 0.6 ms steps of the headline started at 0.70 ms and
    # took more than 20 steps to settle -- 0.67 instead of 0.61 ms per step in a `--steps 20 --warmup 5` run.)
    others = {}
    if not args.no_layers:
        for name in sorted(LAYERS, reverse=True):  # the long one first: it carries the load through the ramp
            if name != args.layer:
                others[name] = Workload(args, name, seeds, rank, world, device)
    wl_strong = None
    if args.scaling == "both" and args.strong_rooms >= world:
        sseeds = strong_seeds()
        if sseeds != seeds:  # (one rank of an 8-rank run with 8 rooms owns exactly its weak-scaling room)
            wl_strong = Workload(args, args.layer, sseeds, rank, world, device)
    layers = None if args.no_layers else {}
    for name, w2 in others.items():
        ms, val, _ = w2.timed(max(args.steps // 2, 3), 2)
        layers[name] = {"ms_per_step": round(ms, 4), "value": round(val, 1), "unit": "points/s", "edges_per_gpu": w2.e_local,
                        "mode": "pipelined" if w2.pipeline else "sequential"}
        if w2.pipeline:  # the same steps strictly one after the other, reported beside the chosen mode
            w2.pipeline = False
            ms_s, val_s, _ = w2.timed(max(args.steps // 2, 3), 2)
            w2.pipeline = True
            layers[name]["sequential_ms_per_step"] = round(ms_s, 4)
            if ms_s < ms:
                layers[name].update(ms_per_step=round(ms_s, 4), value=round(val_s, 1), mode="sequential")

    # ------------------------------------------------------------------ the headline region
    ms_per_step, value, m_total = wl.timed(args.steps, max(args.warmup - 1, 0))
    rank_stats = wl.rank_stats
    ms_sequential = ms_pipelined = None
    headline_mode = "sequential"
    if wl.pipeline:  # the same K steps strictly one after the other (nothing prefetched)
        wl.pipeline = False
        ms_sequential, value_seq, _ = wl.timed(args.steps, 2)
        wl.pipeline = True
        ms_pipelined, headline_mode = ms_per_step, "pipelined"
        if ms_sequential < ms_per_step:
            # e.g. a slow host, or ranks sharing one GPU: the pipelined form is the host-heavier one. Both regions are
            # K timed steps of the same work; the faster one is the headline, the other is reported beside it.
            ms_per_step, value, headline_mode = ms_sequential, value_seq, "sequential"
            rank_stats = wl.rank_stats
    if layers is not None:
        layers[args.layer] = {"ms_per_step": round(ms_per_step, 4), "value": round(value, 1), "unit": "points/s",
                              "edges_per_gpu": wl.e_local, "mode": headline_mode}
        if ms_sequential is not None:
            layers[args.layer]["sequential_ms_per_step"] = round(ms_sequential, 4)

    # ------------------------------------------------------------------ per-op breakdowns and rooflines (rank 0)
    roofline = breakdown = None
    if rank == 0 and not args.no_breakdown:
        roofline, breakdown = wl.breakdown()
        for name, w2 in list(others.items()) + ([(args.layer, wl)] if layers is not None else []):
            rl, bd = (roofline, breakdown) if w2 is wl else w2.breakdown()
            ent = layers[name]
            ent["roofline"] = rl
            ent["conv_ms"] = {"fwd": bd["spatial_conv_fwd"]["ms"], "bwd": bd["spatial_conv_bwd"]["ms"]}
            # conv-only rate: spatial_conv forward + backward with the neighbour list and PDFs cached, as
            # every further layer over the same (level, radius) sees it through ConvolutionBuilder's caches
            ent["conv_only_points_per_s"] = round(w2.P.shape[0] / ((bd["spatial_conv_fwd"]["ms"] + bd["spatial_conv_bwd"]["ms"]) * 1e-3), 1)
            ent["conv_rate"] = {"fwd": bd["spatial_conv_fwd"], "bwd": bd["spatial_conv_bwd"]}
            if "spatial_conv_bf16_rows" in bd:
                ent["bf16_rows"] = bd["spatial_conv_bf16_rows"]
    # ------------------------------------------------------------------ strong scaling: the fixed batch split over the ranks
    strong = None
    if args.scaling == "both" and args.strong_rooms >= world:
        sw = wl_strong if wl_strong is not None else wl
        s_ms, s_val, s_total = sw.timed(max(args.steps // 4, 5) if wl_strong is not None else args.steps, 3)
        strong = {"rooms": args.strong_rooms, "rooms_on_rank0": len(strong_seeds()), "points_total": int(s_total),
                  "ms_per_step": round(s_ms, 4), "value": round(s_val, 1), "unit": "points/s", "scaling": "strong",
                  "mode": "pipelined" if sw.pipeline else "sequential", "rank_stats": sw.rank_stats,
                  "note": "fixed batch of %d rooms split cloud-per-GPU over %d rank(s); speed-up at N ranks = value(N) / "
                          "value(1)" % (args.strong_rooms, world)}
        if rank == 0 and world == 1 and isinstance(roofline, dict) and not args.no_breakdown:
            try:   # the neighbour search on the whole 8-room batch (330 MB of algorithmic bytes): the size at which HBM shows
                roofline["find_neighbors_%drooms" % args.strong_rooms] = sw.find_neighbors_roofline()
            except Exception as ex:
                roofline["find_neighbors_%drooms" % args.strong_rooms] = {"error": repr(ex)[:200]}
        if wl_strong is not None:
            # (no torch.cuda.empty_cache() here: returning gigabytes to the driver idles the GPU for tens of milliseconds,
            # and the headline region below would start at the clocks of an idle chip -- 0.67 instead of 0.61 ms per step
            # when the region is short, `--steps 20 --warmup 5`)
            del wl_strong, sw

    others.clear()
    torch.cuda.empty_cache()

    # ------------------------------------------------------------------ BASELINE.json configurations (rank 0, N == 1)
    configs = None
    if rank == 0 and world == 1 and not args.no_configs:
        configs = {}
        # GPU steps of all configurations first: the CPU legs keep 128 OpenMP threads busy for seconds, and the steps of the
        # small configurations are bound by the host thread that issues them (cfg4 right after cfg3's CPU leg: +15 %)
        pending = []
        for name in ((args.config,) if args.config else ("cfg0", "cfg1", "cfg2", "cfg3", "cfg4")):
            try:
                if args.no_cpu_baseline:
                    configs[name] = run_config(name, device, args, False)
                else:
                    configs[name], cw_ = run_config(name, device, args, "later")
                    pending.append((name, cw_))
            except Exception as ex:
                configs[name] = {"error": repr(ex)}
                log("config %s failed: %r" % (name, ex))
        for name, cw_ in pending:
            config_cpu_leg(configs[name], cw_, name)
        del pending
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ CPU baseline (rank 0, N == 1)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(wl, wl.out)
        except Exception as ex:  # the baseline is informative; never fail the bench on it
            cpu = {"error": repr(ex)}

    if rank == 0:
        fin, fout, combin = LAYERS[args.layer]
        B = len(seeds)
        if args.scaling == "strong":
            par = "fixed batch of %d rooms split cloud-per-GPU over %d rank(s)" % (args.strong_rooms, world)
        else:
            par = "cloud-per-GPU dp%d" % world
        rec = {
            "metric": "MC-convolved points/sec (fwd+bwd), 100k-pt cloud r=0.1",
            "value": round(value, 1), "unit": "points/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if args.scaling == "strong" else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            # the same K steps strictly one after the other, nothing carried over or run ahead: SURVEY 8(d)'s literal
            # definition of the metric, beside the pipelined `value` (every pipelined step still executes all of it)
            "sequential": ({"ms_per_step": round(ms_sequential, 4), "value": round(m_total / (ms_sequential * 1e-3), 1),
                            "unit": "points/s"} if ms_sequential is not None else
                           {"ms_per_step": round(ms_per_step, 4), "value": round(value, 1), "unit": "points/s"}),
            "config": {"workload": "ScanNet-like non-uniform room, %d pts/room, %d room(s) on rank 0, absolute radius %g, "
                                   "KDE window %g, same-level conv %s (Fin=%d, Fout=%d, %s), avg on"
                                   % (args.points, B, args.radius, args.window, args.layer, fin, fout,
                                      "combin" if combin else "depth-wise"),
                       "points_total": int(m_total), "points_per_gpu": int(wl.P.shape[0]), "edges_per_gpu": wl.e_local,
                       "layer": args.layer, "parallelism": par,
                       "headline_mode": headline_mode,
                       "pipeline": ("grid build / neighbour search / KDE of batch k+1 on a side stream under the convolution "
                                    "kernels of batch k (ConvolutionBuilder.prefetch_geometry); every step still runs all of "
                                    "them" if wl.pipeline else None),
                       "pipelined_ms_per_step": (round(ms_pipelined, 4) if ms_pipelined is not None else None),
                       "sequential_ms_per_step": (round(ms_sequential, 4) if ms_sequential is not None else None),
                       "collective_backend": (backend if dist_on else None),
                       "rccl_world_size": (dist.get_world_size() if dist_on else 1),
                       "collectives_warmed_before_warmup": bool(dist_on), "rank_stats": rank_stats},
            "roofline": roofline, "cpu_baseline": cpu, "strong": strong, "layers": layers, "configs": configs,
            "breakdown": breakdown,
        }
        try:  # RCCL prints its version banner through C stdio (NCCL_DEBUG=VERSION): push it out BEFORE the record
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        emit_record(rec, args.details)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
