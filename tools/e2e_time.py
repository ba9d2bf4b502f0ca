"""Wall time of one MCClassS training step (cfg1: 32 clouds x 1024 points, k = 16), with the autograd engine's worker
thread and on the calling thread: python tools/e2e_time.py (on the GPU box)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mcclass_s import MCClassS, synthetic_batch  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
B, n, k = 32, 1024, 16
net = MCClassS(1, B, k, 40, dev)
P, Bi, F, y = synthetic_batch(B, n, 40, rng, dev)
net(P, Bi, F, True)
opt = torch.optim.Adam(net.parameters(), lr=1e-3)


AHEAD = [None]


def step(pipelined=False):
    ahead = AHEAD[0]
    if pipelined:  # the next batch's point hierarchy one step ahead (here: the same batch again)
        AHEAD[0] = net.prefetch_hierarchy(P, Bi)
    logits = net(P, Bi, F, True, prefetched=ahead if pipelined else None)
    loss = torch.nn.functional.cross_entropy(logits, y)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for mt in (True, False):
    torch.autograd.set_multithreading_enabled(mt)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("MCClassS cfg1 (32 x 1024 pts, k=16), autograd multithreading %s: %.2f ms/step, %.0f clouds/s, levels %s"
          % (mt, dt * 1e3, B / dt, [int(p.shape[0]) for p in net.lastHierarchy.points_]))
AHEAD[0] = net.prefetch_hierarchy(P, Bi)
for _ in range(5):
    step(True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step(True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("... with the next batch's hierarchy one step ahead (PointHierarchy.prefetch): %.2f ms/step, %.0f clouds/s" % (dt * 1e3, B / dt))

# ... and two batches ahead: the next batch's hierarchy is complete when a step starts, its geometry and row plans are started
# under this batch's layers (ConvolutionBuilder.prefetch_step through forward(nextHierarchy=...))
ph_cur = net.hierarchy(P, Bi, F)
AFTER = True if os.environ.get("E2E_AFTER", "1") == "1" else None   # the synthetic batch is resident: its hierarchy starts at once
net.convBuilder.hostStepsAhead_ = 1
fut = net.prefetch_hierarchy(P, Bi, after=AFTER)


def deep_step():
    global ph_cur, fut
    ph_nxt = net.hierarchy(P, Bi, F, prefetched=fut)
    fut = net.prefetch_hierarchy(P, Bi, after=AFTER)
    logits = net(P, Bi, F, True, hierarchy=ph_cur, nextHierarchy=ph_nxt)
    loss = torch.nn.functional.cross_entropy(logits, y)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    ph_cur = ph_nxt


for _ in range(5):
    deep_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    deep_step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("... hierarchy two batches ahead + prefetch_step: %.2f ms/step, %.0f clouds/s" % (dt * 1e3, B / dt))
