import sys, os, math
sys.argv=[sys.argv[0]]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)),'perf.py')).read().split('print("n_leaves"')[0])
from grace_hip import sharding
gh.enable_kernel_timing(True)
gh.trace_prepare(s,tree)
for world in (1,2,4,8):
    a,b=sharding.shard_bounds(R,world,0); mr=rays[a:b].contiguous(); out=torch.empty(b-a,dtype=torch.float32,device=dev)
    mn,md=timeit(lambda: gh.trace_cumulative_sph(mr,s,tree,out),15)
    gh.trace_prepare_rays(mr)
    mn2,md2=timeit(lambda: gh.trace_cumulative_sph(mr,s,tree,out),15)
    print("shard 1/%d: call %.3f -> prepared rays %.3f (kernel %.3f)"%(world,md,md2,gh.last_kernel_ms()), flush=True)
    gh.trace_release_rays()
