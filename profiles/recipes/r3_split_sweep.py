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
    cnt = torch.empty(len(sh), dtype=torch.int32, device=dev)
    gh.trace_prepare_rays(sh)
    for K in (4, 2, 1, -1, 2, 4, -1):
        gh.set_packet_split(K)
        a, b = kms(lambda: gh.trace_cumulative_sph(sh, s, tree, out))
        c, e = kms(lambda: gh.trace_hitcounts_sph(sh, s, tree, cnt))
        print("1/%d shard split %2d: cum mean %.4f min %.4f | count mean %.4f min %.4f ms" % (d, K, a, b, c, e), flush=True)
    gh.set_packet_split(-1)
