import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "grace-devel_amd")); sys.path.insert(0, ROOT)
import torch, hashlib
import grace_hip as gh
from bench import make_particles
dev = torch.device("cuda:0")
n = 2_000_000
s = make_particles(n, dev)
lo, hi = gh.min_max_vec4(s); lo[3] = hi[3] = 0.0
tree = gh.Tree(n, 32, device=dev)
gh.build_tree(s, tree, lo[:3], hi[:3])
for side in (512, 200):
    rays, _ = gh.orthogonal_rays_z(side, lo, hi, device=dev)
    for nr in (len(rays), len(rays) // 8):
        r = rays[:nr].contiguous()
        out = torch.empty(nr, dtype=torch.float32, device=dev)
        cnt = torch.empty(nr, dtype=torch.int32, device=dev)
        for split in (-1, 1, 4):
            gh.set_packet_split(split)
            gh.trace_cumulative_sph(r, s, tree, out); gh.trace_hitcounts_sph(r, s, tree, cnt)
            h = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
            hc = hashlib.sha256(cnt.cpu().numpy().tobytes()).hexdigest()[:16]
            print("side %d rays %d split %d: cum %s count %s" % (side, nr, split, h, hc), flush=True)
        gh.set_packet_split(-1)
