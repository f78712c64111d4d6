import sys, os, math, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT,'grace-devel_amd')); sys.path.insert(0, os.path.join(ROOT,'oracle')); sys.path.insert(0, os.path.join(ROOT,'tests'))
import torch, numpy as np, grace_hip as gh
from test_gpu_triangles import heightfield_mesh, _cameras
dev=torch.device('cuda:0')
def timeit(f,reps=5):
    f(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
tris=heightfield_mesh(1024,512)
d=torch.from_numpy(tris).to(dev)
tree=gh.Tree(len(tris),32,device=dev)
bot,top=gh.build_tree_tris(d,tree)
cams,center,up,fovy,length=_cameras(np.array(bot,np.float64),np.array(top,np.float64),50.,1024,1024)
ts=[]; hs=[]
for k,cam in enumerate(cams):
    rays=gh.pinhole_camera_rays(1024,1024,cam,center,up,fovy,length,device=dev)
    cl=torch.empty(len(rays),dtype=torch.int32,device=dev)
    ts.append(timeit(lambda: gh.trace_closest_tri(rays,d,tree,cl)))
    hs.append(hashlib.sha256(cl.cpu().numpy().tobytes()).hexdigest()[:12])
print("config5 cameras %.3f %.3f %.3f ms  hashes %s"%(*ts, " ".join(hs)), flush=True)
gh.trace_status()
