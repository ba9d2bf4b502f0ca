"""Where the MCClassS training step goes: GPU-busy time vs wall, most expensive kernels and host ops."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
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
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5): step()
    torch.cuda.synchronize()
ka = prof.key_averages()
gpu = sum(e.self_device_time_total for e in ka) / 5e3
print("GPU busy per step: %.2f ms" % gpu)
print(ka.table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=50))
print(ka.table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=50))
