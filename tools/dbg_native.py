import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.helpers import make_cloud
import tests.test_gpu_native as T
import mccnn_amd.MCConvModule as mc
only = sys.argv[1:] or None
if only:
    T.LAYERS = [l for l in T.LAYERS if l[0] in only]
pts, bids = make_cloud(3000, 3, 5, "clustered", True)
feats, ogs = {}, {}
cb0, ph0, outs0, grads0, names0 = T._run(mc, False, pts, bids, 3, True, feats, ogs)
sd = {k: v.detach().clone() for k, v in cb0.state_dict().items()}
cb1, ph1, outs1, grads1, names1 = T._run(mc, True, pts, bids, 3, True, feats, ogs, state=sd)
for l, a, b in zip(T.LAYERS, outs0, outs1):
    d = (a.float() - b.float()).abs()
    bad = (d > 1e-4 * a.float().abs().max()).any(dim=1).nonzero().reshape(-1)
    print(l[0], "max diff", float(d.max()), "scale", float(a.float().abs().max()), "bad rows", bad.numel(), bad[:20].tolist())
    if bad.numel():
        r = int(bad[0]); print(" op", a[r].tolist(), "\n nat", b[r].tolist())
for i, (a, b) in enumerate(zip(grads0, grads1)):
    what = T.LAYERS[i][0] + ":featGrad" if i < len(T.LAYERS) else names0[i - len(T.LAYERS)]
    d = (a.float() - b.float()).abs()
    print(what, "max diff", float(d.max()), "scale", float(a.float().abs().max()))
# the oracle's answer for the layers (same parameters, CPU tensors through the builder)
from oracle.oracle import Oracle
from tests.oracle_ops import OracleOps
from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder
oo = OracleOps(Oracle(omp=True))
Pc, Bc = torch.from_numpy(pts), torch.from_numpy(bids)
phc = PointHierarchy(Pc, torch.ones((len(pts), 1)), Bc, [0.1], "PH", 3, True, ops=oo)
cbc = ConvolutionBuilder(KDEWindow=0.25, relativeRadius=True, ops=oo)
cbc.load_state_dict({k: v.cpu() for k, v in sd.items()})
cbc.reset()
for li, (name, lin, lout, fin, fout, combin, radius, bf16) in enumerate(T.LAYERS):
    if bf16:
        continue
    f = feats[name].detach().cpu().clone().requires_grad_(True)
    o = cbc.create_convolution(name, phc, lin, f, fin, radius, phc, lout, combin, fout)
    for tag, got in (("op", outs0[li]), ("nat", outs1[li])):
        d = (got.float().cpu() - o.detach()).abs()
        print(name, tag, "vs oracle: max diff", float(d.max()), "scale", float(o.abs().max()))
