import sys, os, math
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT,'grace-devel_amd'))
import torch, numpy as np, grace_hip as gh
dev=torch.device('cuda:0')
n=10_000_000
g=torch.Generator(device=dev); g.manual_seed(42)
s=torch.empty((n,4),dtype=torch.float32,device=dev)
s[:,:3]=torch.rand((n,3),generator=g,device=dev); s[:,3]=float((3*48/(4*math.pi*n))**(1/3))
u=s.clone()
for rep in range(3):
    s.copy_(u)
    tree=gh.Tree(n,32,device=dev)
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record(); gh.build_tree(s,tree,(0,0,0),(1,1,1)); b.record(); torch.cuda.synchronize()
    print("build_tree %.3f ms  n_leaves %d"%(a.elapsed_time(b), tree.n_leaves))
