import sys, os, shutil
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
shutil.copy(ROOT+"/scratch/"+os.environ.get("STAMPS_LIB","lib_stamps.so"), ROOT+"/grace-devel_amd/lib/libgrace_hip.so")
sys.argv=[sys.argv[0]]
exec(open(os.path.join(ROOT,'profiles','recipes','perf.py')).read().split('print("n_leaves"')[0])
from grace_hip import sharding
gh.trace_prepare(s,tree)
for world,K in ((1,-1),(8,-1)):
    a,b=sharding.shard_bounds(R,world,0); mr=rays[a:b].contiguous(); out=torch.empty(b-a,dtype=torch.float32,device=dev)
    gh.set_packet_split(K)
    print("---- shard 1/%d K %d"%(world,K), file=sys.stderr, flush=True)
    for _ in range(2): gh.trace_cumulative_sph(mr,s,tree,out)
    torch.cuda.synchronize()
