# many random builds vs the oracle tree (wider than the committed fuzz test)
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT,'grace-devel_amd')); sys.path.insert(0, os.path.join(ROOT,'oracle'))
import torch, numpy as np, grace_hip as gh, oracle as O
cuda=torch.device('cuda:0')
bad=0; N=int(sys.argv[1]) if len(sys.argv)>1 else 3000
for seed in range(N):
    rng=np.random.default_rng(900000+seed)
    n=int(rng.choice([2,3,5,17,63,64,65,127,129,255,256,257,300,1000,4097,30000]))
    mpl=int(rng.integers(1,70))
    if n<=mpl: mpl=max(1,n-1)
    s=O.random_real4(n,(0,0,0,0),(1,1,1,0.05),first=int(rng.integers(0,10**6)))
    if rng.random()<0.5:
        k=max(1,n//int(rng.choice([2,10,100]))); s[:,:3]=s[rng.integers(0,k,n),:3]
    use_xor=bool(rng.random()<0.5)
    keys=O.morton_keys30(s,(0,0,0),(1,1,1)); keys_s,ss,_=O.sort_by_key(keys,s); ss=np.ascontiguousarray(ss)
    dl=O.deltas_xor(keys_s) if use_xor else O.deltas_euclid(ss)
    nodes,leaves,root,_=O.albvh(ss,dl,mpl)
    d=torch.from_numpy(ss).to(cuda); tree=gh.Tree(n,mpl,device=cuda)
    if use_xor:
        dk=torch.from_numpy(keys_s.view(np.int32)).to(cuda); dx=torch.empty(n+1,dtype=torch.int32,device=cuda)
        gh.XOR_deltas_sph(dk,dx); gh.ALBVH_sph(d,dx,tree)
    else:
        df=torch.empty(n+1,dtype=torch.float32,device=cuda); gh.euclidean_deltas_sph(d,df); gh.ALBVH_sph(d,df,tree)
    ok=np.array_equal(tree.leaves.cpu().numpy(),leaves) and np.array_equal(tree.nodes.cpu().numpy(),nodes) and int(tree.root_index.item())==root
    if not ok:
        bad+=1; print("MISMATCH",seed,n,mpl,use_xor,flush=True)
    if seed%500==0: print("..",seed,flush=True)
print("done",N,"bad",bad)
