import sys, torch, numpy as np
sys.path.insert(0,'/root/repo')
from mccnn_amd import MCConvModule as M
rng=np.random.default_rng(1)
P=torch.tensor(rng.random((4096,3),dtype=np.float32),device='cuda'); Bi=torch.zeros((4096,1),dtype=torch.int32,device='cuda')
mn,mx=M.compute_aabb(P,Bi,1,True)
keys,idx=M.sort_points_step1(P,Bi,mn,mx,1,0.1,True)
sP,sB,sF,cells=M.sort_points_step2(P,Bi,P.clone(),keys,idx,mn,mx,1,0.1,True)
st,pk=M.find_neighbors(P,Bi,sP,cells,mn,mx,0.1,1,True)
ref=M.compute_pdf(sP,sB,mn,mx,st,pk,0.2,0.1,1,True,mode=0).flatten().cpu().numpy()
out=M.compute_pdf(sP,sB,mn,mx,st,pk,0.2,0.1,1,True,mode=1).flatten().cpu().numpy()
s=st.flatten().cpu().numpy()
r=out/ref
print("ratio quantiles", np.quantile(r,[0,0.01,0.5,0.99,1]))
for row in range(3):
    a,b=s[row],s[row+1]; print(row,b-a, np.round(r[a:b],3))
