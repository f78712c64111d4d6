# extended run of tests/test_gpu_fuzz.py::_one over many seeds (time-bounded)
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT,'grace-devel_amd')); sys.path.insert(0, os.path.join(ROOT,'oracle')); sys.path.insert(0, os.path.join(ROOT,'tests'))
import torch, numpy as np, grace_hip as gh, oracle as O
import test_gpu_fuzz as F
dev=torch.device('cuda:0')
budget=float(sys.argv[1]) if len(sys.argv)>1 else 300
t0=time.time(); n=0; bad=0; seed=100000
while time.time()-t0<budget:
    rng=np.random.default_rng(seed)
    try:
        F._one(gh,O,dev,rng,seed)
    except AssertionError as e:
        bad+=1; print("FAIL seed",seed,str(e)[:300],flush=True)
    finally:
        gh.set_ray_reorder(True); gh.set_packet_split(-1); gh.set_treelet_size(-1); gh.set_packet_width(-1); gh.set_exact_integrals(False)
    n+=1; seed+=1
    if n%200==0: print("..",n,"configs",int(time.time()-t0),"s",flush=True)
print("done",n,"configs, failures",bad)
