# full frame + shards + ragged / jittered orthographic batches (call medians)
import sys, os, math
sys.argv=[sys.argv[0]]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)),'perf.py')).read().split('print("n_leaves"')[0])
from grace_hip import sharding
gh.enable_kernel_timing(True)
gh.trace_prepare(s,tree)
for world in (1,2,4,8):
    a,b=sharding.shard_bounds(R,world,0); mr=rays[a:b].contiguous(); out=torch.empty(b-a,dtype=torch.float32,device=dev)
    mn,md=timeit(lambda: gh.trace_cumulative_sph(mr,s,tree,out),9); print("shard 1/%d: call median %.3f kernel %.3f"%(world,md,gh.last_kernel_ms()), flush=True)
for side2 in (1000, 724):
    r2,_=gh.orthogonal_rays_z(side2,lo,hi,device=dev); r2=r2[:(len(r2)//64)*64].contiguous()
    out=torch.empty(len(r2),dtype=torch.float32,device=dev)
    mn,md=timeit(lambda: gh.trace_cumulative_sph(r2,s,tree,out),9); print("grid %d^2: call median %.3f kernel %.3f -> %.1f Mrays/s"%(side2,md,gh.last_kernel_ms(),len(r2)/md/1e3), flush=True)
lo3=[float(x) for x in lo[:3]]; hi3=[float(x) for x in hi[:3]]
rj=gh.plane_parallel_random_rays(1024,1024,(lo3[0],lo3[1],lo3[2]-0.01),(hi3[0]-lo3[0],0,0),(0,hi3[1]-lo3[1],0),hi3[2]-lo3[2]+0.02,seed=5,device=dev)
out=torch.empty(len(rj),dtype=torch.float32,device=dev)
mn,md=timeit(lambda: gh.trace_cumulative_sph(rj,s,tree,out),9); print("jittered 1024^2: call median %.3f kernel %.3f -> %.1f Mrays/s"%(md,gh.last_kernel_ms(),len(rj)/md/1e3), flush=True)
