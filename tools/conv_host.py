import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mccnn_amd.MCConvBuilder as MB
from mccnn_amd import native
from tests.helpers import make_cloud
torch.autograd.set_multithreading_enabled(False)
pts, bids = make_cloud(300, 1, 5, "uniform", True)
P, Bi = torch.from_numpy(pts).cuda(), torch.from_numpy(bids).cuda()
ph = MB.PointHierarchy(P, torch.ones((len(pts), 1), device="cuda"), Bi, [], "PH", 1)
b = MB.ConvolutionBuilder(KDEWindow=0.2)
for fin, comb, fout in ((64, False, 64), (1, True, 16)):
    f = torch.rand((len(pts), fin), device="cuda", requires_grad=True)
    name = "c%d" % fin
    out = b.create_convolution(name, ph, 0, f, fin, 0.3, outNumFeatures=fout, multiFeatureConv=comb)
    og = torch.rand_like(out)
    params = [p for n, p in b.named_parameters() if n.startswith(name)]
    N = 2000
    for mode in ("fwd_nograd", "fwd", "fwd+bwd"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            if mode == "fwd_nograd":
                with torch.no_grad():
                    out = b.create_convolution(name, ph, 0, f, fin, 0.3, outNumFeatures=fout, multiFeatureConv=comb)
            else:
                out = b.create_convolution(name, ph, 0, f, fin, 0.3, outNumFeatures=fout, multiFeatureConv=comb)
                if mode == "fwd+bwd":
                    torch.autograd.grad([out], [f] + params, [og])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("fin %d %s: host %.1f us per call (with sync %.1f)" % (fin, mode, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
    # host time of the backward call alone (the forward outputs made first, queue drained)
    outs = [b.create_convolution(name, ph, 0, f, fin, 0.3, outNumFeatures=fout, multiFeatureConv=comb) for _ in range(200)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for o in outs:
        torch.autograd.grad([o], [f] + params, [og])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("fin %d backward alone: host %.1f us per call (with sync %.1f)" % (fin, (t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
    import cProfile, pstats
    outs = [b.create_convolution(name, ph, 0, f, fin, 0.3, outNumFeatures=fout, multiFeatureConv=comb) for _ in range(200)]
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for o in outs:
        torch.autograd.grad([o], [f] + params, [og])
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(6)
    # the interleaved loop again, host time of its two halves
    tf = tb = 0.0
    torch.cuda.synchronize()
    for _ in range(500):
        a0 = time.perf_counter()
        out = b.create_convolution(name, ph, 0, f, fin, 0.3, outNumFeatures=fout, multiFeatureConv=comb)
        a1 = time.perf_counter()
        torch.autograd.grad([out], [f] + params, [og])
        a2 = time.perf_counter()
        tf += a1 - a0
        tb += a2 - a1
    torch.cuda.synchronize()
    print("fin %d interleaved: forward %.1f us, backward %.1f us on the host" % (fin, tf / 500 * 1e6, tb / 500 * 1e6))
