"""Sort micro-benchmark: 10^7 30-bit keys + 16-byte payload in place (the build's sort), also u64/63 bits,
keys only, and a clustered key set.  Prints ms (min / median of reps) and checks against torch.sort."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "grace-devel_amd")); sys.path.insert(0, ROOT)
import torch
import grace_hip as gh
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
def run(name, kmaster, bits, vwords, reps=12, check=True):
    vmaster = torch.randint(0, 2**31 - 1, (n, vwords), dtype=torch.int32, device=dev, generator=g) if vwords else None
    k = kmaster.clone(); v = vmaster.clone() if vwords else None
    ts = []
    for r in range(reps):
        k.copy_(kmaster)
        if vwords: v.copy_(vmaster)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); gh.sort_by_key(k, v, 0, bits); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ok = ""
    if check:
        ku = kmaster.to(torch.int64) if kmaster.dtype == torch.int32 else kmaster
        sk, si = torch.sort(ku, stable=True)
        good = bool((k.to(torch.int64) == sk).all())
        if vwords: good = good and bool((v == vmaster[si]).all())
        ok = "OK" if good else "MISMATCH"
    ts = sorted(ts[2:])
    print("%-40s min %.4f  med %.4f ms  %s" % (name, ts[0], ts[len(ts) // 2], ok), flush=True)
k30 = torch.randint(0, 2**30, (n,), dtype=torch.int32, device=dev, generator=g)
run("u32 30 bits + 16 B", k30, 30, 4)
if os.environ.get("ONLY_FIRST"): sys.exit(0)
run("u32 30 bits keys only", k30, 30, 0)
run("u32 30 bits + 32 B", k30, 30, 8)
k63 = torch.randint(0, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)
run("u64 63 bits + 16 B", k63, 63, 4)
# clustered: half the keys inside 1/4096 of the key range
kc = k30.clone(); kc[: n // 2] = (kc[: n // 2] & 0x3FFFF) | (0x155 << 18)
run("u32 30 bits clustered + 16 B", kc, 30, 4)
kd = torch.randint(0, 50, (n,), dtype=torch.int32, device=dev, generator=g)
run("u32 30 bits, 50 distinct + 16 B", kd, 30, 4)
