import sys, torch, numpy as np
sys.path.insert(0,'/root/repo')
import bench, argparse
from mccnn_amd import MCConvModule as M
ap = argparse.Namespace(points=100000, radius=0.1, window=0.2, layer='1to64', rooms_per_gpu=1, steps=20, warmup=5, scaling='weak', strong_rooms=8, gpus=1, no_cpu_baseline=True, no_breakdown=True, no_layers=True, no_pipeline=True)
wl = bench.Workload(ap, '1to64', [20180601], 0, 1, torch.device('cuda',0))
P,Bi,B=wl.P,wl.Bi,wl.B; mn,mx=wl.ph.aabbMin_,wl.ph.aabbMax_
keys,idx=M.sort_points_step1(P,Bi,mn,mx,B,0.1,False)
sP,sB,sF,cells=M.sort_points_step2(P,Bi,wl.F.detach(),keys,idx,mn,mx,B,0.1,False)
st,pk=M.find_neighbors(P,Bi,sP,cells,mn,mx,0.1,B,False)
ref=M.compute_pdf(sP,sB,mn,mx,st,pk,0.2,0.1,B,False,mode=0)
for mode in (2,1):
    out=M.compute_pdf(sP,sB,mn,mx,st,pk,0.2,0.1,B,False,mode=mode)
    rel=((out-ref).abs()/ref.abs()).max().item()
    for _ in range(10): M.compute_pdf(sP,sB,mn,mx,st,pk,0.2,0.1,B,False,mode=mode)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): M.compute_pdf(sP,sB,mn,mx,st,pk,0.2,0.1,B,False,mode=mode)
    e1.record(); torch.cuda.synchronize()
    print("mode",mode,"max per-value rel err vs mode0 %.2e"%rel,"ms %.4f"%(e0.elapsed_time(e1)/50))
