"""Quick A/B of the frame kernel: 10^7 particles, 1024^2 rays; prints kernel ms (library events) for
fast / exact integrals and hit counts, plus the 1/8 shard."""
import sys, os, math, time
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
cnt = torch.empty(len(rays), dtype=torch.int32, device=dev)
gh.set_cache_validation(False); gh.trace_prepare(s, tree)
gh.enable_kernel_timing(True)
def kms(fn, reps=10):
    fn(); fn(); v = []
    for _ in range(reps):
        fn(); v.append(gh.last_kernel_ms())
    return sum(v) / len(v), min(v)
res = {}
gh.trace_prepare_rays(rays)
res["frame fast"] = kms(lambda: gh.trace_cumulative_sph(rays, s, tree, out))
gh.set_exact_integrals(True)
res["frame exact"] = kms(lambda: gh.trace_cumulative_sph(rays, s, tree, out), 5)
gh.set_exact_integrals(False)
res["frame counts"] = kms(lambda: gh.trace_hitcounts_sph(rays, s, tree, cnt), 5)
for d in (2, 4, 8):
    sh = rays[: len(rays) // d].contiguous()
    o2 = out[: len(sh)]
    gh.trace_prepare_rays(sh)
    res["1/%d shard fast" % d] = kms(lambda: gh.trace_cumulative_sph(sh, s, tree, o2))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): gh.trace_cumulative_sph(sh, s, tree, o2)
    torch.cuda.synchronize(); res["1/%d shard call (trusted)" % d] = (1e3 * (time.perf_counter() - t0) / 20, 0)
for k, (a, b) in res.items():
    print("%-28s mean %.4f ms  min %.4f ms" % (k, a, b))
