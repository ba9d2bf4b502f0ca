import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from tests.helpers import make_room
from mccnn_amd.MCConvBuilder import PointHierarchy
import mccnn_amd.MCConvModule as M
P = torch.from_numpy(make_room(100000, 20180601)).cuda()
Bi = torch.zeros((100000,1), dtype=torch.int32, device='cuda')
F = torch.ones((100000,1), device='cuda')
for it in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    ph = PointHierarchy(P, F, Bi, [0.1,0.2,0.4,0.8], "PH", 1, False)
    torch.cuda.synchronize(); print("hierarchy ms", (time.perf_counter()-t0)*1e3, [int(p.shape[0]) for p in ph.points_])
mn,mx = ph.aabbMin_, ph.aabbMax_
k,i = M.sort_points_step1(P,Bi,mn,mx,1,0.1,False); sP,sB,sF,c = M.sort_points_step2(P,Bi,F,k,i,mn,mx,1,0.1,False)
for it in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    out = M.poisson_sampling(sP,sB,c,mn,mx,0.1,1,False)
    torch.cuda.synchronize(); print("poisson 100k r=0.1 ms", (time.perf_counter()-t0)*1e3, out[0].shape[0])
