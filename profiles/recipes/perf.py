import sys, os, math, time
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT,'grace-devel_amd'))
import torch, numpy as np, grace_hip as gh
dev=torch.device('cuda:0')
n=int(sys.argv[1]) if len(sys.argv)>1 else 10_000_000
side=int(sys.argv[2]) if len(sys.argv)>2 else 1024
mpl=int(sys.argv[3]) if len(sys.argv)>3 else 32
g=torch.Generator(device=dev); g.manual_seed(42)
s=torch.empty((n,4),dtype=torch.float32,device=dev)
s[:,:3]=torch.rand((n,3),generator=g,device=dev)
s[:,3]=float((3*48/(4*math.pi*n))**(1/3))
lo,hi=gh.min_max_vec4(s); lo[3]=hi[3]=0
tree=gh.Tree(n,mpl,device=dev); gh.build_tree(s,tree,lo[:3],hi[:3])
rays,_=gh.orthogonal_rays_z(side,lo,hi,device=dev)
R=len(rays)
hc=torch.empty(R,dtype=torch.int32,device=dev); cu=torch.empty(R,dtype=torch.float32,device=dev)
def timeit(f,reps=5):
    f(); torch.cuda.synchronize()
    ts=[]
    for _ in range(reps):
        a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts), sorted(ts)[len(ts)//2]
print("n_leaves",tree.n_leaves)
for T in (-1,):
  gh.set_treelet_size(T); print("treelet",T)
  for name,f in [("hitcounts",lambda: gh.trace_hitcounts_sph(rays,s,tree,hc)),
               ("cumulative",lambda: gh.trace_cumulative_sph(rays,s,tree,cu))]:
    mn,md=timeit(f)
    print("   ","%-12s min %.3f ms  median %.3f ms  -> %.1f Mrays/s"%(name,mn,md,R/md/1e3))
gh.trace_status()
