import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "grace-devel_amd")); sys.path.insert(0, ROOT)
import torch
import grace_hip as gh
from bench import make_particles
dev = torch.device("cuda:0")
n = 10_000_000
s = make_particles(n, dev)
lo, hi = gh.min_max_vec4(s); lo[3] = hi[3] = 0.0
tree = gh.Tree(n, 32, device=dev)
gh.build_tree(s, tree, lo[:3], hi[:3])
rays, _ = gh.orthogonal_rays_z(1024, lo, hi, device=dev)
out = torch.empty(len(rays), dtype=torch.float32, device=dev)
gh.set_cache_auto(False)
for _ in range(8): gh.trace_cumulative_sph(rays, s, tree, out)
torch.cuda.synchronize()
