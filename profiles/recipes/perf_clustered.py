# clustered scene (background + Gaussian clumps, h from the analytic local density): rate per ray and per hit
import sys, os, math
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT,'grace-devel_amd'))
import torch, numpy as np, grace_hip as gh
dev=torch.device('cuda:0')
def timeit(f,reps=5):
    f(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
n=10_000_000
g=torch.Generator(device=dev); g.manual_seed(11)
for frac_clump, n_clumps in ((0.0,0),(0.5,200),(0.9,50)):
    nb=int(n*(1-frac_clump)); nc=n-nb
    pos=torch.rand((nb,3),generator=g,device=dev)
    dens=torch.full((nb,),float(nb),device=dev)
    if nc:
        centres=torch.rand((n_clumps,3),generator=g,device=dev)*0.8+0.1
        sig=10**(torch.rand(n_clumps,generator=g,device=dev)*1.5-3.0)     # 1e-3 .. 3e-2
        which=torch.randint(0,n_clumps,(nc,),generator=g,device=dev)
        p=centres[which]+torch.randn((nc,3),generator=g,device=dev)*sig[which,None]
        per=nc/n_clumps
        r2=((p-centres[which])**2).sum(1)/sig[which]**2
        d=per*torch.exp(-0.5*r2)/((2*math.pi)**1.5*sig[which]**3)+nb
        pos=torch.cat([pos,p.clamp(0,1)]); dens=torch.cat([dens,d])
    h=(3*48/(4*math.pi*dens))**(1/3)
    s=torch.cat([pos,h[:,None]],1).float().contiguous()
    lo,hi=gh.min_max_vec4(s); lo[3]=hi[3]=0
    tree=gh.Tree(n,32,device=dev); gh.build_tree(s,tree,lo[:3],hi[:3])
    rays,_=gh.orthogonal_rays_z(1024,lo,hi,device=dev); R=len(rays)
    hc=torch.empty(R,dtype=torch.int32,device=dev); cu=torch.empty(R,dtype=torch.float32,device=dev)
    gh.trace_prepare(s,tree)
    if os.environ.get('TREELET'): gh.set_treelet_size(int(os.environ['TREELET']))
    t0=timeit(lambda: gh.trace_hitcounts_sph(rays,s,tree,hc)); hits=hc.double().sum().item()
    t1=timeit(lambda: gh.trace_cumulative_sph(rays,s,tree,cu))
    print("lattice",gh.last_lattice())
    print("clump fraction %.1f: h %.2e..%.2e  hits/ray mean %.0f max %d | hitcounts %.2f ms, cumulative %.2f ms = %.0f Mrays/s, %.1f G hits/s"%(frac_clump,h.min().item(),h.max().item(),hits/R,hc.max().item(),t0,t1,R/t1/1e3,hits/t1/1e6),flush=True)
    gh.trace_release() if hasattr(gh,'trace_release') else None
    del tree,s,rays
