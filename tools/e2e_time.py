import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mcclass_s import MCClassS, synthetic_batch
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
B, n, k = 32, 1024, 16
net = MCClassS(1, B, k, 40, dev)
P, Bi, F, y = synthetic_batch(B, n, 40, rng, dev)
net(P, Bi, F, True)
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
def step():
    logits = net(P, Bi, F, True)
    loss = torch.nn.functional.cross_entropy(logits, y)
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("MCClassS cfg1 (32 x 1024 pts, k=16): %.2f ms/step, %.0f clouds/s, levels %s" % (dt * 1e3, B / dt, [int(p.shape[0]) for p in net.lastHierarchy.points_]))
