"""Round 3: rank 0's shard of the 1024^2 frame (10^7 particles) for 1, 2, 4, 8 ranks -- the three
flavours of the trace_cumulative_sph call: stateless with the library's validated cache in its
steady state (the default), cold (caching off: every record re-derived), trusted (validation off,
caches pinned) -- and the kernel's own time."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "grace-devel_amd")); sys.path.insert(0, ROOT)
import torch
import grace_hip as gh
from grace_hip import sharding
from bench import make_particles
dev = torch.device("cuda:0")
n = 10_000_000
s = make_particles(n, dev)
lo, hi = gh.min_max_vec4(s); lo[3] = hi[3] = 0.0
tree = gh.Tree(n, 32, device=dev)
gh.build_tree(s, tree, lo[:3], hi[:3])
rays, _ = gh.orthogonal_rays_z(1024, lo, hi, device=dev)
gh.enable_kernel_timing(True)

def call_ms(fn, reps=30):
    for _ in range(4): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps

base = None
for world in (1, 2, 4, 8):
    a, b = sharding.shard_bounds(len(rays), world, 0)
    mr = rays[a:b].contiguous(); out = torch.empty(b - a, dtype=torch.float32, device=dev)
    f = lambda: gh.trace_cumulative_sph(mr, s, tree, out)
    stateless = call_ms(f); kern = gh.last_kernel_ms()
    gh.set_cache_auto(False); cold = call_ms(f, 10); gh.set_cache_auto(True)
    gh.set_cache_validation(False); gh.trace_prepare(s, tree); gh.trace_prepare_rays(mr)
    trusted = call_ms(f)
    gh.trace_release(); gh.trace_release_rays(); gh.set_cache_validation(True)
    base = base or (stateless, cold, trusted)
    print("shard 1/%d (%7d rays): stateless %.3f ms (x%.2f) | cold %.3f ms (x%.2f) | trusted %.3f ms (x%.2f) | kernel %.3f ms"
          % (world, b - a, stateless, base[0] / stateless, cold, base[1] / cold, trusted, base[2] / trusted, kern), flush=True)
gh.trace_status()
