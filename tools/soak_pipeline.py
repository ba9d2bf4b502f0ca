"""Soak test of the pipelined loop: thousands of steps over batches of different sizes in random order, no host
synchronisation inside the loop (mismatches against the per-batch reference are counted ON the GPU), so that a
lifetime / ordering bug between the two streams would show up as a wrong forward output or gradient."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.helpers import make_cloud
from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder
torch.autograd.set_multithreading_enabled(False)
STEPS = int(os.environ.get("SOAK_STEPS", "3000"))
rng = np.random.default_rng(1)
batches = []
for n_per, B, seed, kind in ((6000, 2, 51, "uniform"), (15000, 2, 52, "clustered"), (3000, 3, 53, "uniform"),
                             (12000, 2, 54, "clustered"), (20000, 1, 55, "uniform"), (9000, 4, 56, "uniform")):
    pts, bids = make_cloud(n_per, B, seed, kind, True)
    P = torch.from_numpy(pts).cuda(); Bi = torch.from_numpy(bids).cuda()
    F = torch.from_numpy(rng.random((len(pts), 1), dtype=np.float32)).cuda().requires_grad_(True)
    og = torch.from_numpy(rng.random((len(pts), 16), dtype=np.float32)).cuda()
    batches.append((PointHierarchy(P, F, Bi, [], "PH", B, False), F, og))
torch.manual_seed(5)
builder = ConvolutionBuilder(KDEWindow=0.2, relativeRadius=False)
def conv(ph, F, og):
    F.grad = None
    for p in builder.parameters():
        p.grad = None
    out = builder.create_convolution("Conv", ph, 0, F, 1, 0.08, outNumFeatures=16, multiFeatureConv=True)
    out.backward(og)
    return out
refs = []
for ph, F, og in batches:
    builder.reset()
    out = conv(ph, F, og)
    refs.append((out.detach().clone(), F.grad.clone(), [p.grad.clone() for p in builder.parameters()]))
order = rng.integers(0, len(batches), STEPS)
bad_out = torch.zeros((), dtype=torch.int64, device="cuda")
worst_g = torch.zeros((), dtype=torch.float32, device="cuda")
builder.reset()
builder.prefetch_geometry(batches[order[0]][0], 0, 0.08)
torch.cuda.synchronize()
t0 = time.perf_counter()
for step in range(STEPS):
    ph, F, og = batches[order[step]]
    builder.reset()
    out = conv(ph, F, og)
    if step + 1 < STEPS:
        builder.prefetch_geometry(batches[order[step + 1]][0], 0, 0.08)
    r = refs[order[step]]
    bad_out += (out.detach() != r[0]).sum()
    worst_g = torch.maximum(worst_g, (F.grad - r[1]).abs().max() / r[1].abs().max())
    for g, rg in zip([p.grad for p in builder.parameters()], r[2]):
        worst_g = torch.maximum(worst_g, (g - rg).abs().max() / rg.abs().max().clamp_min(1e-30))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("soak: %d pipelined steps in %.1f s, forward mismatches %d, worst relative gradient deviation %.2e, "
      "memory %.0f MB" % (STEPS, dt, int(bad_out.item()), float(worst_g.item()), torch.cuda.max_memory_allocated() / 1e6))
assert int(bad_out.item()) == 0 and float(worst_g.item()) < 1e-4
