"""GPU-side cost of a hipEventRecord between two kernels of one stream (does the record flush the caches?), of a
cross-stream wait, and of record_stream bookkeeping: wall time of 300 iterations, queue drained at the end."""
import time, torch
dev = torch.device("cuda", 0)
big = torch.empty(32 * 1024 * 1024, dtype=torch.float32, device=dev)   # 128 MB written per iteration
small = torch.zeros(64, device=dev)
side = torch.cuda.Stream()
def run(body, n=300):
    for _ in range(20):
        body()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        body()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
def plain():
    big.fill_(1.0); small.add_(1.0)
def with_event():
    big.fill_(1.0); e = torch.cuda.Event(); e.record(); small.add_(1.0)
def with_timing_event():
    big.fill_(1.0); e = torch.cuda.Event(enable_timing=True); e.record(); small.add_(1.0)
def with_3_events():
    big.fill_(1.0)
    for _ in range(3):
        e = torch.cuda.Event(); e.record()
    small.add_(1.0)
def cross_wait_ready():   # wait on an event of the side stream that completed long ago
    big.fill_(1.0); torch.cuda.current_stream().wait_event(done); small.add_(1.0)
with torch.cuda.stream(side):
    small2 = torch.zeros(64, device=dev); small2.add_(1.0)
done = torch.cuda.Event(); done.record(side)
torch.cuda.synchronize()
base = run(plain)
print("us per iteration: plain %.1f, +event %.1f, +timing event %.1f, +3 events %.1f, +wait on a finished event %.1f" % (
    base, run(with_event), run(with_timing_event), run(with_3_events), run(cross_wait_ready)))
