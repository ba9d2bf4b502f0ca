"""Soak test of a whole network's pipelined loop on the native executor: thousands of MCClassH-graph steps (three-level
hierarchy, 10 convolutions over 7 neighbour lists, depth-wise layers with row plans) over batches of DIFFERENT sizes in
random order, with everything that runs ahead switched on -- the next batch's PointHierarchy on its helper thread, the
learned geometry prefetch on side streams, the row plans / transposed lists on the third helper thread -- and no host
synchronisation inside the loop. Every step's outputs and gradients are compared ON the GPU with the per-batch reference
computed with all of it switched off; capacity guesses overflow whenever a larger batch follows a smaller one.
    SOAK_STEPS=2000 python tools/soak_network.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder  # noqa: E402
from mccnn_amd.workloads import CONFIGS, config_points  # noqa: E402

torch.cuda.set_device(0)
torch.autograd.set_multithreading_enabled(os.environ.get("AUTOGRAD_MT", "0") == "1")   # (AUTOGRAD_MT=1: the backward passes on the engine's device thread)
STEPS = int(os.environ.get("SOAK_STEPS", "2000"))
cfg = CONFIGS[os.environ.get("SOAK_CFG", "cfg2")]   # cfg4: absolute radii (box extent read back per hierarchy), 17 layers
dev = torch.device("cuda", 0)
rng = np.random.default_rng(7)


VARY_GRAPH = False                     # switched on after the references (SOAK_VARY_GRAPH=1)
SKIP_BWD = False                       # (SOAK_SKIP_BWD=1)
vrng = np.random.default_rng(11)


class Batch:
    def __init__(self, clouds, points, seed):
        c = cfg._replace(clouds=clouds, points=points, seed=seed)
        p, b, self.B = config_points(c)
        self.P, self.Bi = torch.from_numpy(p).to(dev), torch.from_numpy(b).to(dev)
        # input feature rows that differ from point to point: the level rows gathered WITH a prefetched hierarchy
        # (PointHierarchy.prefetch(features=), SOAK_FEATS=0 switches it off) are compared with the inline gathers below
        gF = torch.Generator(device="cpu").manual_seed(seed)
        self.F0 = torch.rand((len(p), 3), generator=gF).to(dev)
        self.feats = self.ogs = None
        self.ref_level_feats = None

    def hierarchy(self, prefetched=None):
        return PointHierarchy(self.P, self.F0, self.Bi, list(cfg.hierarchy), "PH", self.B, cfg.relative, prefetched=prefetched)

    def request(self):
        return PointHierarchy.prefetch(self.P, self.Bi, list(cfg.hierarchy), self.B, cfg.relative,
                                       after=(True if os.environ.get("SOAK_HIER_AFTER", "1") == "1" else None),
                                       features=(self.F0 if os.environ.get("SOAK_FEATS", "1") == "1" else None))

    def rows(self, ph):
        if self.feats is None:
            self.feats, self.ogs = [], []
            for ci, c in enumerate(cfg.convs):
                g = torch.Generator(device="cpu").manual_seed(100 + ci)
                n, m = int(ph.points_[c.lin].shape[0]), int(ph.points_[c.lout].shape[0])
                self.feats.append((2 * torch.rand((n, c.fin), generator=g) - 1).to(dev).requires_grad_(True))
                self.ogs.append((2 * torch.rand((m, c.fout if c.combin else c.fin), generator=g) - 1).to(dev))


def step(builder, batch, prefetched=None, ready=None, then=None):
    """ready: this batch's hierarchy, constructed a step ago; then: called after reset() (the deep pipeline starts the next
    batch's geometry there)."""
    builder.reset()
    if then is not None:
        then()
    ph = ready if ready is not None else batch.hierarchy(prefetched)
    batch.rows(ph)
    if batch.ref_level_feats is None:     # (the reference pass: built inline, nothing prefetched)
        batch.ref_level_feats = [f.detach().clone() for f in ph.features_]
    else:
        global feat_bad
        for f, r in zip(ph.features_, batch.ref_level_feats):
            feat_bad += (f.detach() != r).sum()
    use = list(range(len(cfg.convs)))
    if VARY_GRAPH and vrng.random() < 0.4:      # a step whose graph differs: some layers missing, the rest in another order
        use = [ci for ci in use if vrng.random() < 0.7]
        vrng.shuffle(use)
        use = use or [0]
    outs = [None] * len(cfg.convs)
    for ci in use:
        c = cfg.convs[ci]
        outs[ci] = builder.create_convolution(c.name, ph, c.lin, batch.feats[ci], c.fin, c.radius, ph, c.lout, c.combin, c.fout, c.window)
    if SKIP_BWD and vrng.random() < 0.3:         # a step without a backward pass (pieces prebuilt for it are never consumed)
        return outs, [None] * (len(batch.feats) + len(list(builder.parameters())))
    grads = torch.autograd.grad([outs[ci] for ci in use], batch.feats + list(builder.parameters()), [batch.ogs[ci] for ci in use],
                                allow_unused=True)
    return outs, grads


feat_bad = torch.zeros((), dtype=torch.int64, device=dev)
if cfg.cloud_kind == "room":
    SHAPES = ((1, 100000, 20180601), (1, 60000, 7), (2, 40000, 11), (1, 80000, 19), (1, 30000, 23))
else:
    SHAPES = ((8, 4096, 43), (5, 3000, 44), (12, 2048, 45), (6, 4096, 46), (3, 8192, 47), (10, 1024, 48))
batches = [Batch(c, p, s) for c, p, s in SHAPES]
torch.manual_seed(3)
builder = ConvolutionBuilder(KDEWindow=0.25, relativeRadius=cfg.relative)
step(builder, batches[0])  # creates the variables
# references: nothing runs ahead
builder.geoPrefetch_ = False
refs = []
for b in batches:
    outs, grads = step(builder, b)
    refs.append(([o.detach().clone() for o in outs], [None if g is None else g.clone() for g in grads]))
builder.geoPrefetch_ = True
VARY_GRAPH = os.environ.get("SOAK_VARY_GRAPH") == "1"
SKIP_BWD = os.environ.get("SOAK_SKIP_BWD") == "1"
if os.environ.get("SOAK_LAG"):   # the steps issued one at a time (ConvolutionBuilder.hostStepsAhead_ = 0)
    builder.hostStepsAhead_ = int(os.environ["SOAK_LAG"])
order = rng.integers(0, len(batches), STEPS)
bad = torch.zeros((), dtype=torch.int64, device=dev)
worst = torch.zeros((), dtype=torch.float32, device=dev)
TRACE = torch.zeros((STEPS, len(cfg.convs)), dtype=torch.int64, device=dev) if os.environ.get("SOAK_TRACE") else None
DEEP = os.environ.get("SOAK_DEEP", "0") == "1"   # hierarchy two batches ahead + ConvolutionBuilder.prefetch_step for the next
ADOPT_FIRST = os.environ.get("SOAK_ADOPT_FIRST", "0") == "1"
ahead = batches[order[0]].request()
if DEEP:
    ready = batches[order[0]].hierarchy(ahead)
    ahead = batches[order[1]].request()
torch.cuda.synchronize()
torch.cuda.reset_peak_memory_stats()
m0 = torch.cuda.memory_allocated()
t0 = time.perf_counter()
for s in range(STEPS):
    b = batches[order[s]]
    if DEEP:
        state = {}

        def start_next():
            if s + 1 < STEPS:
                state["nxt"] = batches[order[s + 1]].hierarchy(ahead)
                builder.prefetch_step(state["nxt"])
        if ADOPT_FIRST:   # examples/mcclass_s.py's order: next hierarchy adopted and the one after it requested BEFORE reset()'s wait
            if s + 1 < STEPS:
                state["nxt"] = batches[order[s + 1]].hierarchy(ahead)
            ahead = batches[order[s + 2]].request() if s + 2 < STEPS else None
            outs, grads = step(builder, b, ready=ready,
                               then=(lambda: builder.prefetch_step(state["nxt"])) if "nxt" in state else None)
            ready = state.get("nxt")
        else:
            outs, grads = step(builder, b, ready=ready, then=start_next)
            ready = state.get("nxt")
            ahead = batches[order[s + 2]].request() if s + 2 < STEPS else None
    else:
        cur, ahead = ahead, (batches[order[s + 1]].request() if s + 1 < STEPS else None)
        outs, grads = step(builder, b, cur)
    r_out, r_grad = refs[order[s]]
    if os.environ.get("SOAK_DEBUG"):   # (synchronises every step: which layer of which batch differs first, and its list)
        for ci, (o, r) in enumerate(zip(outs, r_out)):
            if o is not None and not torch.equal(o.detach(), r):
                c = cfg.convs[ci]
                print("step %d batch %d (previous %d): layer %s (levels %d -> %d, radius %g) differs: %d of %d values" % (
                    s, order[s], order[s - 1] if s else -1, c.name, c.lin, c.lout, c.radius, int((o.detach() != r).sum()), r.numel()))
                for key, g in builder.cacheGeo_.items():
                    print("   ", key, "n", g.n, "m", g.m, "e", g.e, "e_cap", g.e_cap, "side", getattr(g.core, "side", None))
                sys.exit(1)
    for ci, (o, r) in enumerate(zip(outs, r_out)):
        if o is None:
            continue
        nb_ = (o.detach() != r).sum()
        bad += nb_
        if TRACE is not None:
            TRACE[s, ci] = nb_
    full = all(o is not None for o in outs)
    for g, r in zip(grads, r_grad):
        if g is not None and full:     # (a step with layers missing leaves other gradients: only the outputs are compared)
            worst = torch.maximum(worst, (g - r).abs().max() / r.abs().max().clamp_min(1e-30))
    if os.environ.get("SOAK_MEM") and s % 5000 == 4999:
        print("  step %d: %.1f MB allocated" % (s + 1, torch.cuda.memory_allocated() / 1e6), flush=True)
torch.cuda.synchronize()
if TRACE is not None:
    nz = TRACE.nonzero().cpu().numpy()
    for st_, ci in nz[:12]:
        c = cfg.convs[ci]
        print("  mismatch: step %d batch %d (previous %d, next %d) layer %s (levels %d -> %d, r %g): %d values" % (
            st_, order[st_], order[st_ - 1] if st_ else -1, order[st_ + 1] if st_ + 1 < STEPS else -1, c.name, c.lin, c.lout, c.radius,
            int(TRACE[st_, ci])))
dt = time.perf_counter() - t0
print("soak_network%s: %d steps in %.1f s (%.2f ms/step), forward mismatches %d, worst relative gradient deviation %.2e, "
      "memory now %.0f MB (start %.0f), peak %.0f MB; level feature rows: %d mismatches" % (
          " (deep)" if DEEP else "", STEPS, dt, dt / STEPS * 1e3, int(bad.item()), float(worst.item()),
          torch.cuda.memory_allocated() / 1e6, m0 / 1e6, torch.cuda.max_memory_allocated() / 1e6, int(feat_bad.item())))
assert int(bad.item()) == 0 and float(worst.item()) < 1e-4 and int(feat_bad.item()) == 0
