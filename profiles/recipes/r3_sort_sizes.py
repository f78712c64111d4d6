import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "grace-devel_amd")); sys.path.insert(0, ROOT)
import torch
import grace_hip as gh
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
def run(n, bits, words, perm, reps=14):
    km = torch.randint(0, 2**bits if bits < 31 else 2**31 - 1, (n,), dtype=torch.int32, device=dev, generator=g)
    vm = torch.randint(0, 2**31 - 1, (n, words), dtype=torch.int32, device=dev, generator=g) if words else None
    k = km.clone(); v = vm.clone() if words else None
    ts = []
    for r in range(reps):
        k.copy_(km)
        if words: v.copy_(vm)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); gh.sort_by_key(k, v, 0, bits, want_perm=perm); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[3:])
    print("n %9d bits %2d words %d perm %d : min %.4f med %.4f ms" % (n, bits, words, perm, ts[0], ts[len(ts)//2]), flush=True)
for n in (1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22):
    run(n, 30, 4, False)
for n in (1 << 18, 1 << 20, 1 << 22):
    run(n, 16, 0, True)
    run(n, 24, 0, True)
    run(n, 30, 0, False)
