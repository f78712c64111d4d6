# python prof_cfg.py <2|3> <cum|hits|count> [reps]
import sys, os, math
cfg=int(sys.argv[1]); mode=sys.argv[2]; reps=int(sys.argv[3]) if len(sys.argv)>3 else 3
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT,'grace-devel_amd')); sys.path.insert(0, os.path.join(ROOT,'oracle'))
import torch, numpy as np, grace_hip as gh, oracle as O
dev=torch.device('cuda:0')
if cfg==2:
    n=1_000_000; R=100_000
    s=torch.from_numpy(O.random_real4(n,(0,0,0,0),(1,1,1,0.1))).to(dev)
    tree=gh.Tree(n,32,device=dev); gh.build_tree(s,tree,(0,0,0),(1,1,1))
    rays=gh.uniform_random_rays(R,(0.5,0.5,0.5),2.0,seed=1234,device=dev)
else:
    n_side=128; n=n_side**3
    g=torch.Generator(device=dev); g.manual_seed(42)
    grid=torch.stack(torch.meshgrid(*[torch.arange(n_side,device=dev)]*3,indexing="ij"),-1).reshape(-1,3).float()
    pos=(grid+torch.rand((n,3),generator=g,device=dev))/n_side
    h=(3*48/(4*math.pi*n))**(1/3)
    s=torch.cat([pos,torch.full((n,1),h,device=dev)],1).contiguous()
    lo,hi=gh.min_max_vec4(s)
    tree=gh.Tree(n,32,device=dev); gh.build_tree(s,tree,lo[:3],hi[:3])
    centre=(lo[:3]+hi[:3])/2; length=float(np.linalg.norm(hi[:3]-lo[:3]))
    rays=gh.healpix_rays(64,centre,length,device=dev)
R=len(rays)
cu=torch.empty(R,dtype=torch.float32,device=dev); hc=torch.empty(R,dtype=torch.int32,device=dev)
gh.trace_prepare(s,tree)
if os.environ.get('TREELET'): gh.set_treelet_size(int(os.environ['TREELET']))
def timeit(f,reps=5):
    f(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
f={"cum":lambda: gh.trace_cumulative_sph(rays,s,tree,cu),"count":lambda: gh.trace_hitcounts_sph(rays,s,tree,hc),"hits":lambda: gh.trace_sph(rays,s,tree)}[mode]
gh.enable_kernel_timing(True)
print("config %d %s: median %.3f ms (last kernel %.3f ms)"%(cfg,mode,timeit(f,reps),gh.last_kernel_ms()))
gh.trace_status()
