import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT,'grace-devel_amd')); sys.path.insert(0, os.path.join(ROOT,'oracle'))
import torch, numpy as np, grace_hip as gh, oracle as O
dev=torch.device('cuda:0')
def timeit(f,reps=5):
    f(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
g=torch.Generator(device=dev); g.manual_seed(5)
for n, rmax in ((1_000_000, 0.02), (10_000_000, 0.008)):
    s=torch.rand((n,4),generator=g,device=dev); s[:,3]*=rmax; s=s.contiguous()
    tree=gh.Tree(n,32,device=dev); gh.build_tree(s,tree,(0,0,0),(1,1,1))
    for R, kind in ((262144,'random'),(262144,'random_short'),(65536,'two_origins')):
        rays=torch.zeros((R,7),device=dev)
        d=torch.randn((R,3),generator=g,device=dev); d/=d.norm(dim=1,keepdim=True)
        rays[:,:3]=d
        if kind=='two_origins':
            rays[:,3:6]=torch.where(torch.arange(R,device=dev)[:,None]%2==0, torch.tensor([0.3,0.3,0.3],device=dev), torch.tensor([0.7,0.6,0.5],device=dev))
            rays[:,6]=2.0
        else:
            rays[:,3:6]=torch.rand((R,3),generator=g,device=dev)
            rays[:,6]=2.0 if kind=='random' else 0.1
        rays=rays.contiguous()
        cu=torch.empty(R,dtype=torch.float32,device=dev); hc=torch.empty(R,dtype=torch.int32,device=dev)
        t1=timeit(lambda: gh.trace_cumulative_sph(rays,s,tree,cu)); t0=timeit(lambda: gh.trace_hitcounts_sph(rays,s,tree,hc))
        print("n %8d %-13s R %7d: cumulative %.3f ms, hitcounts %.3f ms, mean hits %.1f, sum %.6e"%(n,kind,R,t1,t0,hc.float().mean().item(),cu.double().sum().item()),flush=True)
    del s, tree
# pencils with few rays (wide packets)
n=10_000_000
s=torch.rand((n,4),generator=g,device=dev); s[:,3]*=0.008; s=s.contiguous()
tree=gh.Tree(n,32,device=dev); gh.build_tree(s,tree,(0,0,0),(1,1,1))
for R in (64, 1024, 12288, 196608):
    rays=gh.uniform_random_rays(R,(0.5,0.5,0.5),2.0,seed=7,device=dev)
    cu=torch.empty(R,dtype=torch.float32,device=dev); hc=torch.empty(R,dtype=torch.int32,device=dev)
    t1=timeit(lambda: gh.trace_cumulative_sph(rays,s,tree,cu)); t0=timeit(lambda: gh.trace_hitcounts_sph(rays,s,tree,hc))
    print("n %8d pencil        R %7d: cumulative %.3f ms, hitcounts %.3f ms, mean hits %.1f, sum %.6e"%(n,R,t1,t0,hc.float().mean().item(),cu.double().sum().item()),flush=True)
