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
gh.set_cache_validation(False); gh.trace_prepare(s, tree)
gh.enable_kernel_timing(True)
def kms(fn, reps=8):
    fn(); fn(); v = []
    for _ in range(reps):
        fn(); v.append(gh.last_kernel_ms())
    return sum(v) / len(v), min(v)
for d in (8, 4, 2, 1):
    sh = rays[: len(rays) // d].contiguous()
    out = torch.empty(len(sh), dtype=torch.float32, device=dev)
    gh.trace_prepare_rays(sh)
    for T in (-1, 16384, 32768, 65536, 131072, 262144, 524288, 2097152):
        gh.set_treelet_size(T)
        a, b = kms(lambda: gh.trace_cumulative_sph(sh, s, tree, out))
        print("1/%d shard treelet %6d: mean %.4f min %.4f ms" % (d, T, a, b), flush=True)
    gh.set_treelet_size(-1)
