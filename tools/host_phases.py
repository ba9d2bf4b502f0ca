"""Host time of the phases of one pipelined bench step (perf_counter around reset / forward / backward / prefetch),
GPU running asynchronously: which phase keeps the main queue waiting."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.autograd.set_multithreading_enabled(False)
ap = argparse.Namespace(points=100000, radius=0.1, window=0.2, layer='1to64', rooms_per_gpu=1, steps=20, warmup=5,
                        scaling='weak', strong_rooms=8, gpus=1, no_cpu_baseline=True, no_breakdown=True, no_layers=True,
                        no_pipeline=False)
wl = bench.Workload(ap, '1to64', [20180601], 0, 1, torch.device('cuda', 0))
b, a = wl.builder, wl.args
if os.environ.get("NO_MAIN_WAIT") == "1":   # timing experiment only (results unsafe): the main queue does not wait for the side queue
    _real = torch.cuda.Stream.wait_event
    def _patched(self, ev):
        if b.sideStream_ is not None and self.cuda_stream == b.sideStream_.cuda_stream:
            return _real(self, ev)
    torch.cuda.Stream.wait_event = _patched
acc = {"reset": 0.0, "grads_none": 0.0, "forward": 0.0, "backward": 0.0, "prefetch": 0.0}
evs = []
side_evs = []
late = []
def step(rec):
    t0 = time.perf_counter()
    b.reset()
    t1 = time.perf_counter()
    if rec:
        if evs:
            late.append(evs[-1].query())   # True: backward k had already finished when the host came out of reset(k+1)
        e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)   # main queue: next forward may start
    wl.F.grad = None
    for p in b.parameters():
        p.grad = None
    t2 = time.perf_counter()
    out = b.create_convolution("Conv", wl.ph, 0, wl.F, wl.fin, a.radius, outNumFeatures=wl.fout, multiFeatureConv=wl.combin, KDEWindow=a.window)
    t3 = time.perf_counter()
    out.backward(wl.OG)
    if rec:
        e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)   # main queue: backward done
    t4 = time.perf_counter()
    b.prefetch_geometry(wl.ph, 0, a.radius, KDEWindow=a.window, transposed=not wl.combin)
    if rec:
        e = torch.cuda.Event(enable_timing=True); e.record(b.sideStream_); side_evs.append(e)   # side queue: geometry k+1 done
    t5 = time.perf_counter()
    if rec:
        for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[k] += v
for _ in range(50):
    step(False)
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for _ in range(N):
    step(True)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / N * 1e3
print("ms/step %.4f; host phases (ms): " % tot + ", ".join("%s %.3f" % (k, v / N * 1e3) for k, v in acc.items()) + "; sum %.3f" % (sum(acc.values()) / N * 1e3))

# evs = [fwd_ok_0, bwd_done_0, fwd_ok_1, bwd_done_1, ...]: main-queue time from the end of backward k to the point where
# forward k+1 may start (includes the GPU-side wait for the side stream's event)
gaps = [evs[2 * k + 1].elapsed_time(evs[2 * k + 2]) * 1e3 for k in range(N - 1)]
busy = [evs[2 * k].elapsed_time(evs[2 * k + 1]) * 1e3 for k in range(N)]
import statistics
print("main queue: forward+backward kernels %.1f us (median), idle between backward k and forward k+1 %.1f us (median), %.1f (mean)" % (
    statistics.median(busy), statistics.median(gaps), statistics.mean(gaps)))

# positive: the side queue finishes the geometry of batch k+1 AFTER the main queue finished backward k (the next forward waits for it)
lag = [evs[2 * k + 1].elapsed_time(side_evs[k]) * 1e3 for k in range(N - 1)]
print("side queue done minus main backward done: median %.1f us, mean %.1f, min %.1f, max %.1f" % (
    statistics.median(lag), statistics.mean(lag), min(lag), max(lag)))
print("host came out of reset() AFTER the GPU had finished the previous backward in %d of %d steps" % (sum(late), len(late)))
