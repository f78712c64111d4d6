import sys, os, math
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT,'grace-devel_amd')); sys.path.insert(0, os.path.join(ROOT,'oracle')); sys.path.insert(0, os.path.join(ROOT,'tests'))
import torch, numpy as np, grace_hip as gh, oracle as O
dev=torch.device('cuda:0')
def timeit(f,reps=5):
    f(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
TS=(512,2048,4096,8192,16384,32768,65536)
# config 2
n=1_000_000; R=100_000
s=torch.from_numpy(O.random_real4(n,(0,0,0,0),(1,1,1,0.1))).to(dev)
tree=gh.Tree(n,32,device=dev); gh.build_tree(s,tree,(0,0,0),(1,1,1))
rays=gh.uniform_random_rays(R,(0.5,0.5,0.5),2.0,seed=1234,device=dev)
cu=torch.empty(R,dtype=torch.float32,device=dev); hc=torch.empty(R,dtype=torch.int32,device=dev)
for T in TS:
    gh.set_treelet_size(T)
    print("config2 T %5d cumulative %.3f hitcounts %.3f trace_sph %.3f"%(T,timeit(lambda: gh.trace_cumulative_sph(rays,s,tree,cu)),timeit(lambda: gh.trace_hitcounts_sph(rays,s,tree,hc)),timeit(lambda: gh.trace_sph(rays,s,tree),2)), flush=True)
# config 3
n_side=128; n=n_side**3
g=torch.Generator(device=dev); g.manual_seed(42)
grid=torch.stack(torch.meshgrid(*[torch.arange(n_side,device=dev)]*3,indexing="ij"),-1).reshape(-1,3).float()
pos=(grid+torch.rand((n,3),generator=g,device=dev))/n_side
h=(3*48/(4*math.pi*n))**(1/3)
s3=torch.cat([pos,torch.full((n,1),h,device=dev)],1).contiguous()
lo,hi=gh.min_max_vec4(s3)
t3=gh.Tree(n,32,device=dev); gh.build_tree(s3,t3,lo[:3],hi[:3])
centre=(lo[:3]+hi[:3])/2; length=float(np.linalg.norm(hi[:3]-lo[:3]))
r3=gh.healpix_rays(64,centre,length,device=dev)
o3=torch.empty(len(r3),dtype=torch.float32,device=dev)
for T in TS:
    gh.set_treelet_size(T)
    print("config3 T %5d cumulative %.3f trace_sph %.3f"%(T,timeit(lambda: gh.trace_cumulative_sph(r3,s3,t3,o3)),timeit(lambda: gh.trace_sph(r3,s3,t3))), flush=True)
# config 5
from test_gpu_triangles import heightfield_mesh, _cameras
tris=heightfield_mesh(1024,512)
d=torch.from_numpy(tris).to(dev)
tree=gh.Tree(len(tris),32,device=dev)
bot,top=gh.build_tree_tris(d,tree)
cams,center,up,fovy,length=_cameras(np.array(bot,np.float64),np.array(top,np.float64),50.,1024,1024)
for T in TS:
    gh.set_treelet_size(T)
    ts=[]
    for k,cam in enumerate(cams):
        rays=gh.pinhole_camera_rays(1024,1024,cam,center,up,fovy,length,device=dev)
        cl=torch.empty(len(rays),dtype=torch.int32,device=dev)
        ts.append(timeit(lambda: gh.trace_closest_tri(rays,d,tree,cl)))
    print("config5 T %5d cameras %.3f %.3f %.3f"%(T,*ts), flush=True)
gh.trace_status()
