import sys, os, math
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT,'grace-devel_amd'))
import torch, numpy as np, grace_hip as gh
dev=torch.device('cuda:0')
def timeit(f,reps=3):
    f(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
n=1_000_000; R=100_000
g=torch.Generator(device=dev); g.manual_seed(3)
s=torch.rand((n,4),generator=g,device=dev,dtype=torch.float64); s[:,3]*=0.1
tree=gh.Tree(n,32,device=dev); gh.build_tree_d4(s,tree,(0,0,0),(1,1,1))
rays=gh.uniform_random_rays(R,(0.5,0.5,0.5),2.0,seed=1234,device=dev)
hc=torch.empty(R,dtype=torch.int32,device=dev); cu=torch.empty(R,dtype=torch.float64,device=dev)
t=timeit(lambda: gh.trace_hitcounts_d4(rays,s,tree,hc)); print("d4 hitcounts %.3f ms (%.2f Mrays/s), mean hits %.0f"%(t,R/t/1e3,hc.float().mean().item()))
t=timeit(lambda: gh.trace_cumulative_d4(rays,s,tree,cu)); print("d4 cumulative %.3f ms (%.2f Mrays/s)"%(t,R/t/1e3))
sf=s.float().contiguous(); tf=gh.Tree(n,32,device=dev); gh.build_tree(sf,tf,(0,0,0),(1,1,1))
cf=torch.empty(R,dtype=torch.float32,device=dev)
t=timeit(lambda: gh.trace_cumulative_sph(rays,sf,tf,cf)); print("f4 cumulative %.3f ms"%t)
t=timeit(lambda: gh.trace_sph_d4(rays[:32*200].contiguous(),s,tree)); print("d4 trace_sph (6400 rays) %.3f ms"%t)
# config 4 scene in double: 1e7 particles, 1024^2 orthographic rays
n=10_000_000
g.manual_seed(42)
s4=torch.empty((n,4),dtype=torch.float32,device=dev); s4[:,:3]=torch.rand((n,3),generator=g,device=dev); s4[:,3]=float((3*48/(4*math.pi*n))**(1/3))
lo,hi=gh.min_max_vec4(s4); lo[3]=hi[3]=0
sd=s4.double().contiguous(); del s4
td=gh.Tree(n,32,device=dev); gh.build_tree_d4(sd,td,lo[:3],hi[:3])
r4,_=gh.orthogonal_rays_z(1024,lo,hi,device=dev)
cd=torch.empty(len(r4),dtype=torch.float64,device=dev)
t=timeit(lambda: gh.trace_cumulative_d4(r4,sd,td,cd)); print("config4 d4 cumulative %.3f ms (%.1f Mrays/s)"%(t,len(r4)/t/1e3))
gh.trace_status()
