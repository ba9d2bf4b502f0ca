"""Where a pipelined step of a BASELINE configuration spends its time: host time of the phases of ConfigWorkload.step (deep
pipeline: hierarchy two batches ahead + prefetch_step) and, from events on the main queue, the span of the convolution
chain against the step period.   python tools/step_phases.py cfg4 [lag]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from mccnn_amd.workloads import CONFIGS

torch.cuda.set_device(0)
torch.autograd.set_multithreading_enabled(False)
name = sys.argv[1]
lag = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cw = bench.ConfigWorkload(CONFIGS[name], torch.device("cuda", 0))
assert cw.set_pipeline(True, geometry=True)
cw.builder.hostStepsAhead_ = (lag - 1) if lag else None
ph_t = {k: 0.0 for k in ("reset", "adopt", "request", "prefetch", "fwd", "bwd")}
ev = []
N = 60
NOCONV = os.environ.get("NOCONV") == "1"


GQ = []     # main-queue events: start of reset() / after reset() / after the hierarchy's adoption
HOLD = []   # (debugging, HOLD_GEOS=n: the geometries of the last n steps are kept alive -- is a fault tied to their release?)


def step(rec):
    if os.environ.get("HOLD_GEOS"):
        HOLD.append(list(cw.builder.cacheGeo_.values()) + [v[0] for v in cw.builder.prefetchedGeo_.values()])
        del HOLD[:-int(os.environ["HOLD_GEOS"])]
    t = [time.perf_counter()]
    if os.environ.get("REORDER") == "1":   # A/B: adopt the next hierarchy and request the one after it BEFORE reset()'s wait
        ph = cw.ph = cw.ready_ph
        nxt = cw.hierarchy(cw.next_ph)
        cw.request_next()
        cw.builder.reset(); t.append(time.perf_counter()); t.append(time.perf_counter()); t.append(time.perf_counter())
    else:
        gA = torch.cuda.Event(enable_timing=True); gA.record()
        cw.builder.reset(); t.append(time.perf_counter())
        gB = torch.cuda.Event(enable_timing=True); gB.record()
        ph = cw.ph = cw.ready_ph
        nxt = cw.hierarchy(cw.next_ph); t.append(time.perf_counter())
        gC = torch.cuda.Event(enable_timing=True); gC.record()
        cw.request_next(); t.append(time.perf_counter())
        if rec:
            GQ.append((gA, gB, gC))
    cw.builder.prefetch_step(nxt); t.append(time.perf_counter())
    cw.ready_ph = nxt
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    if NOCONV and rec:   # the side chains alone: hierarchy two ahead + the next batch's geometry, no layer consumes them
        t.append(time.perf_counter()); t.append(time.perf_counter())
    else:
        outs = [cw.conv(ph, ci) for ci in range(len(cw.cfg.convs))]; t.append(time.perf_counter())
        cw.grads = torch.autograd.grad(outs, cw.feats + cw.params, cw.ogs, allow_unused=True); t.append(time.perf_counter())
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    if rec:
        for k, a, b in zip(ph_t, t[:-1], t[1:]):
            ph_t[k] += b - a
        ev.append((e0, e1))


if os.environ.get("MAIN_STREAM") == "1":   # A/B: the loop on a stream of its own instead of the device's default (null) stream
    _s = torch.cuda.Stream()
    _ctx = torch.cuda.stream(_s)
    _ctx.__enter__()
for _ in range(10):
    step(False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    step(True)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
el = time.perf_counter() - t0
span = sum(a.elapsed_time(b) for a, b in ev) / N
if GQ and len(GQ) == len(ev):   # where the main QUEUE spends the time between two convolution chains (event records and waits
    k = len(ev) - 1             # cost queue time too: a freed block that carries record_stream() is an event record)
    gaps = [sum(ev[i][1].elapsed_time(GQ[i + 1][0]) for i in range(k)) / k, sum(g[0].elapsed_time(g[1]) for g in GQ) / len(GQ),
            sum(g[1].elapsed_time(g[2]) for g in GQ) / len(GQ), sum(GQ[i][2].elapsed_time(ev[i][0]) for i in range(len(ev))) / len(ev)]
    print("   main queue between two chains, ms: end of backward -> reset() %.3f, across reset() %.3f, across the adoption %.3f, request + prefetch_step -> first layer %.3f" % tuple(gaps))
period = ev[0][0].elapsed_time(ev[-1][0]) / (N - 1)
print("%s lag %d: %.3f ms/step (host issue %.3f); host phases ms: %s" % (name, lag, el / N * 1e3, t_issue / N * 1e3,
      ", ".join("%s %.3f" % (k, v / N * 1e3) for k, v in ph_t.items())))
try:
    from mccnn_amd import native as _nat
    if _nat._EXT is not None:
        dt = _nat._EXT.debug_times(False)
        print("   extension, us per call: " + ", ".join("%s %.1f (x%d/step)" % (k, v[0] / max(v[1], 1) / 1e3, round(v[1] / (N + 10))) for k, v in dt.items()))
except Exception as ex:
    print("   (no extension census: %r)" % (ex,))
print("   main queue: convolution chain spans %.3f ms of a %.3f ms period (first conv launch to end of backward)" % (span, period))
