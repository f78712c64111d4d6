#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the SPH column-density trace (trace_cumulative_sph
semantics) on the project_gadget workload -- 10^7 particles, ONE 1024^2 orthographic image
(BASELINE.json configs[3], the configuration the metric is quoted on; it fits one GPU).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed / RCCL when N > 1): every rank builds the same BVH
from the same seeded particles (the tree is replicated), traces its own contiguous shard of
the job's ray batch, and the per-ray integrals are all-gathered over RCCL (the only
collective; 4 B/ray).  A "step" = one STATELESS call trace_cumulative_sph(rays, spheres, tree, out)
of the whole ray batch -- the call a drop-in user of the reference makes -- (+ the gather when
N > 1), with particles, tree and rays resident in HBM.  The library caches what a call derives
from its inputs alone (pre-pass records, ray coherence order) once it has seen the same arrays
twice, and validates the cache against the arrays' current contents on every call (a signature
pass on the device); the steady state the timed loop measures therefore includes that validation.
Also reported: `cold_call_ms` (caching off: every call re-derives everything) and
`value_trusted` (validation off: the caller promises not to modify the arrays -- round 2's
"prepared" number).

  --scaling strong (default): the job is the single 1024^2 frame of configs[3], cut into N
      contiguous ray shards -- BASELINE.json's "rays sharded over 8xMI355X".
  --scaling weak: N frames of the 1024^2 image (frame r = the grid shifted by a fixed
      sub-pixel offset), one frame per rank; an extra, never the headline.

Rank 0 prints ONE JSON line.  The oracle is used only by the cpu_baseline leg (rank 0,
N = 1) as the thing timed on the host cores, never in the GPU path.

roofline: the traversal is NOT HBM-bound (its working set is cache-resident; measured HBM
traffic is a few % of peak) -- its ceiling is VALU issue.  `achieved` / `peak` are vector
wave-instructions per second: the kernel's SQ_INSTS_VALU per launch (rocprofv3 PMC pass
committed under profiles/, keyed by the sha256 of csrc/trace_kernel.hpp it was collected on; null
when that no longer matches) over the kernel duration measured live in THIS run with HIP
events, against 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 fp32 instruction.
"""
import argparse
import hashlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "grace-devel_amd"))

import numpy as np
import torch

# /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0            # HBM3E spec
N_SIMDS = 256 * 4
MAX_CLOCK_GHZ = 2.4
VALU_PEAK_GINST = N_SIMDS * MAX_CLOCK_GHZ / 2.0   # v_fma_f32 wave64: 2 cycles per SIMD-32
MIN_VALU_PER_HIT = 10   # sub sub mul fma | sqrt mul cvt shl | LDS | fma fma (csrc/trace_kernel.hpp, test-free rounds of fat spheres)
PMC_FILE = os.path.join(ROOT, "profiles", "r03_pmc_trace.json")
TRACE_SRC = os.path.join(ROOT, "grace-devel_amd", "csrc", "trace_kernel.hpp")   # the kernel's source


def make_particles(n, device, seed=42):
    """Synthetic SPH snapshot: n particles uniform in the unit box, smoothing length from
    the ~48-neighbour rule h = (3*48 / (4 pi n))^(1/3)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    s = torch.empty((n, 4), dtype=torch.float32, device=device)
    s[:, :3] = torch.rand((n, 3), generator=g, device=device, dtype=torch.float32)
    s[:, 3] = float((3.0 * 48.0 / (4.0 * math.pi * n)) ** (1.0 / 3.0))
    return s


def _cores():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(spheres_host, rays_host, tree_host, seconds_target=10.0):
    """Host-core baselines, timed in this run on bounded samples (SURVEY.md 8d / BASELINE.md):

    value -- the hot path itself on the CPU: the oracle's restatement of trace_kernel
        (bintree_trace.cuh:52-197; packets of 32 consecutive rays sharing a stack, as a warp
        does) + OnHit_sphere_cumulate, OpenMP over packets on all cores, over the SAME
        particles and the tree the GPU built.  kind = "port".
    tree_traversal -- the reference's own host path (tests/tree_traversal/tree_traversal.cu:
        65-79): OpenMP brute-force sphere_hit over every (ray, sphere) pair, no BVH, on its
        own geometry (config 2: 10^6 spheres, isotropic rays); pair tests/s and implied rays/s.
    morton_key -- the tests/morton_key loop (generic/morton.h:32-42) over the particles, one
        thread (as in the reference) and all threads.
    """
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    O.build()
    cores = _cores()
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    nodes, leaves, root = tree_host
    n_rays = len(rays_host)

    def runs_of_32(count):
        """`count` packets of 32 consecutive rays, evenly spaced over the image."""
        starts = (np.linspace(0, n_rays // 32 - 1, count).astype(np.int64)) * 32
        return rays_host[(starts[:, None] + np.arange(32)[None, :]).reshape(-1)]

    # -- the BVH walk (probe, then ~seconds_target of work)
    probe = runs_of_32(cores)
    t0 = time.perf_counter(); O.trace(probe, spheres_host, nodes, leaves, root, width=32, mode=1)
    per_packet = max((time.perf_counter() - t0) / cores, 1e-6)   # one packet per core
    n_pk = int(min(n_rays // 32, max(cores, seconds_target / per_packet * cores)))
    n_pk = max(cores, (n_pk // cores) * cores)
    sample = runs_of_32(n_pk)
    t0 = time.perf_counter(); O.trace(sample, spheres_host, nodes, leaves, root, width=32, mode=1)
    walk_s = time.perf_counter() - t0
    walk_rate = len(sample) / walk_s

    # -- tree_traversal: brute force, config 2 geometry (10^6 spheres r in U[0, 0.1), isotropic rays)
    n2 = 1_000_000
    s2 = O.random_real4(n2, (0, 0, 0, 0), (1, 1, 1, 0.1))
    rng = np.random.default_rng(1234)
    d = rng.normal(size=(4 * cores, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    r2 = np.zeros((len(d), 7), np.float32); r2[:, :3] = d; r2[:, 3:6] = 0.5; r2[:, 6] = 2.0
    t0 = time.perf_counter(); O.brute_hitcounts(r2, s2); per_ray = (time.perf_counter() - t0) / len(r2)
    n_bf = max(cores, int(min(32000, 0.5 * seconds_target / max(per_ray, 1e-9))) // cores * cores)
    d = rng.normal(size=(n_bf, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    r2 = np.zeros((n_bf, 7), np.float32); r2[:, :3] = d; r2[:, 3:6] = 0.5; r2[:, 6] = 2.0
    t0 = time.perf_counter(); O.brute_hitcounts(r2, s2); bf_s = time.perf_counter() - t0

    # -- morton_key loop over the particles
    nk = min(len(spheres_host), 10_000_000)
    bot = spheres_host[:nk, :3].min(axis=0); top = spheres_host[:nk, :3].max(axis=0)
    keys = np.empty(nk, np.uint32)
    O.morton_keys30(spheres_host[:nk], bot, top, all_threads=True, out=keys)      # touch pages
    t0 = time.perf_counter(); O.morton_keys30(spheres_host[:nk], bot, top, out=keys); k1 = time.perf_counter() - t0
    t0 = time.perf_counter(); O.morton_keys30(spheres_host[:nk], bot, top, all_threads=True, out=keys)
    kn = time.perf_counter() - t0
    return {
        "value": walk_rate / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
        "algorithm": "BVH walk: the oracle's restatement of trace_kernel (32-ray packets, one "
                     "stack per packet) + OnHit_sphere_cumulate, OpenMP over packets",
        "sample": "%d packets of 32 consecutive rays (evenly spaced over the %d-ray image), same "
                  "%d particles and the GPU-built tree, %.1f s wall"
                  % (n_pk, n_rays, len(spheres_host), walk_s),
        "tree_traversal": {
            "algorithm": "brute_force all pairs (the reference's host path has no BVH): "
                         "tests/tree_traversal/tree_traversal.cu:65-79",
            "sample": "%d isotropic rays x %d spheres (config 2 geometry), %.1f s wall" % (n_bf, n2, bf_s),
            "pair_tests_per_s": n_bf * n2 / bf_s, "implied_Mrays_per_s_at_1e6_spheres": n_bf / bf_s / 1e6,
            "cores": cores},
        "morton_key": {"keys": nk, "Mkeys_per_s_1_thread": nk / k1 / 1e6,
                       "Mkeys_per_s_all_threads": nk / kn / 1e6, "cores": cores},
    }


def load_pmc(kernel_name):
    """Counter values of `kernel_name` from the committed rocprofv3 PMC passes, or (None, why)
    if they were collected on a different csrc/trace.hip than the one this library was built
    from."""
    try:
        pmc = json.load(open(PMC_FILE))
    except Exception as e:
        return None, "no PMC file (%s)" % e
    sha = hashlib.sha256(open(TRACE_SRC, "rb").read()).hexdigest()
    if pmc.get("trace_kernel_sha256") != sha:
        return None, "stale: %s was collected on trace_kernel.hpp %s, this is %s" % (
            os.path.relpath(PMC_FILE, ROOT), str(pmc.get("trace_kernel_sha256"))[:12], sha[:12])
    k = pmc.get("kernels", {}).get(kernel_name)
    if not k:
        return None, "kernel %s not in the PMC file" % kernel_name
    return dict(k, source=os.path.relpath(PMC_FILE, ROOT), trace_kernel_sha256=sha[:16],
                collected_at_commit=pmc.get("collected_at_commit")), None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)   # ~1 s timed region at N = 1
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--particles", type=int, default=10_000_000)
    ap.add_argument("--side", type=int, default=1024)
    ap.add_argument("--no-cache", action="store_true",
                    help="grace_trace_set_cache_auto(0): every call re-derives pre-pass records and "
                         "ray order (the cold call) -- the timed loop then measures that")
    ap.add_argument("--max-per-leaf", type=int, default=32)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--trusted", action="store_true",
                    help="grace_trace_set_cache_validation(0) + grace_trace_prepare_*: cached records "
                         "are used on the caller's promise, no signature pass (round 2's prepared call)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # One GPU per rank.  GRACE_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs
    # than ranks (ranks then share devices and collectives go through host memory).
    backend = os.environ.get("GRACE_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(n_dev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def all_reduce(t, op=None):
        """In place; host staging under the gloo rehearsal backend."""
        kw = {} if op is None else {"op": op}
        if backend == "nccl":
            dist.all_reduce(t, **kw)
        else:
            h = t.cpu()
            dist.all_reduce(h, **kw)
            t.copy_(h)

    import grace_hip as gh  # raises if libgrace_hip.so is missing: no fallback

    # ---- build (replicated on every rank; timed separately, not part of a step) --------
    n = args.particles
    spheres = make_particles(n, device)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    lo, hi = gh.min_max_vec4(spheres)
    lo[3] = hi[3] = 0.0
    phases = {}
    unsorted = spheres.clone()
    # Built twice: the first build grows the library's workspace (hipMalloc), the second is
    # the steady-state cost that is reported.  Both start from the same unsorted particles.
    for attempt in range(2):
        spheres.copy_(unsorted)
        tree = gh.Tree(n, args.max_per_leaf, device=device)
        keys = torch.empty(n, dtype=torch.int32, device=device)
        deltas = torch.empty(n + 1, dtype=torch.float32, device=device)
        torch.cuda.synchronize()
        e = [ev() for _ in range(5)]
        e[0].record(); gh.morton_keys_sph(spheres, keys, lo[:3], hi[:3])
        e[1].record(); gh.sort_by_key(keys, spheres, 0, 30)
        e[2].record(); gh.euclidean_deltas_sph(spheres, deltas)
        e[3].record(); gh.ALBVH_sph(spheres, deltas, tree)
        e[4].record(); torch.cuda.synchronize()
        for i, name in enumerate(["morton_ms", "sort_ms", "deltas_ms", "albvh_ms"]):
            phases[name] = round(e[i].elapsed_time(e[i + 1]), 4)
    del unsorted
    del keys, deltas

    # ---- rays: this rank's contiguous shard (a multiple of 64) of the job's ray batch -----
    from grace_hip import sharding
    rays, area = gh.orthogonal_rays_z(args.side, lo, hi, device=device)
    frame_rays = len(rays)
    frames = world if args.scaling == "weak" else 1
    n_rays = frames * frame_rays
    per = sharding.shard_size(n_rays, world)
    r0, r1 = sharding.shard_bounds(n_rays, world, rank)
    if frames > 1:
        # shard == frame `rank` (side^2 is a multiple of 64): shift the grid by this frame's
        # sub-pixel offset (R2 low-discrepancy sequence, |offset| <= 1/4 pixel: stays in the box)
        assert per == frame_rays and r0 == rank * frame_rays
        fx = ((rank * 0.7548776662466927) % 1.0 - 0.5) * 0.5 if rank else 0.0
        fy = ((rank * 0.5698402909980532) % 1.0 - 0.5) * 0.5 if rank else 0.0
        my_rays = rays.clone()
        my_rays[:, 3] += fx * float(hi[0] - lo[0]) / args.side
        my_rays[:, 4] += fy * float(hi[1] - lo[1]) / args.side
    else:
        my_rays = rays[r0:r1].contiguous()
    my_out = torch.zeros(per, dtype=torch.float32, device=device)
    image = my_out

    # ---- algorithmic bytes, counted per ray by the instrumented walk ------
    stats = gh.trace_stats(my_rays, spheres, tree).to(torch.int64).sum(dim=0)
    if world > 1:
        all_reduce(stats)
    nodes_v, leaves_v, tested, hits = [int(x) for x in stats.tolist()]
    alg_bytes_total = 28 * n_rays + 64 * nodes_v + 16 * leaves_v + 16 * tested + 4 * n_rays
    gh.trace_status()

    def trace_once():
        gh.trace_cumulative_sph(my_rays, spheres, tree, my_out[: r1 - r0])

    # A step = this rank's trace + the all-gather of the 4 B/ray results.  Two output buffers:
    # the gather of step k (RCCL's own stream) overlaps the trace of step k + 1; every gather
    # has finished when the timed region ends (GatherPipeline.drain + synchronize).
    pipe = sharding.GatherPipeline(per, world, n_rays, dist, device)
    step_no = [0]

    def step():
        k = step_no[0]; step_no[0] += 1
        buf = pipe.buffer(k)
        gh.trace_cumulative_sph(my_rays, spheres, tree, buf[: r1 - r0])
        return pipe.gather(k)

    def timed_calls(reps):
        trace_once(); trace_once(); trace_once(); torch.cuda.synchronize()    # reach the steady state
        t1 = time.perf_counter()
        for _ in range(reps):
            trace_once()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t1) / reps

    # For the record (never `value`): the cold call -- caching off, every call derives the
    # per-sphere records, node spans, cluster boxes and the ray coherence order afresh -- and the
    # trusted call -- caches pinned, validation off (the caller's promise; round 2's headline).
    gh.set_cache_auto(False)
    cold_ms = timed_calls(5)
    gh.set_cache_auto(True)
    gh.set_cache_validation(False)
    gh.trace_prepare(spheres, tree)
    gh.trace_prepare_rays(my_rays)
    trusted_ms = timed_calls(20)
    gh.trace_release(); gh.trace_release_rays()
    gh.set_cache_validation(True)
    if args.no_cache:
        gh.set_cache_auto(False)
    if args.trusted:
        gh.set_cache_validation(False)
        gh.trace_prepare(spheres, tree)
        gh.trace_prepare_rays(my_rays)

    for _ in range(args.warmup):
        image = step()
    pipe.drain()
    # ---- the timed region: EXACTLY `steps` steps, nothing else in it, launches asynchronous --
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        image = step()
    pipe.drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    # ---- kernel and call durations: HIP events, after the timed region ----------------------
    # In-library events sit directly around the traversal kernel on its stream
    # (grace_trace_enable_timing); torch events around the whole call add the signature pass that
    # validates the cached records (or, cold, the pre-pass and the ray coherence pass).
    gh.enable_kernel_timing(True)
    k_rep = max(3, min(args.steps, 20))
    trace_ms, call_ev = [], []
    for _ in range(k_rep):
        a, b = ev(), ev()
        a.record(); trace_once(); b.record()
        trace_ms.append(gh.last_kernel_ms())
        call_ev.append((a, b))
    torch.cuda.synchronize()
    kern_ms = sum(trace_ms) / len(trace_ms)
    call_ms = sum(a.elapsed_time(b) for a, b in call_ev) / len(call_ev)
    gh.enable_kernel_timing(False)
    t = torch.tensor([elapsed, kern_ms, call_ms], dtype=torch.float64, device=device)
    if world > 1:
        all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kern_ms, call_ms = float(t[0]), float(t[1]), float(t[2])
    gh.trace_status()

    # For the record (not `value`): the same step with the reference's per-hit arithmetic bit for
    # bit (grace_trace_set_exact_integrals(1)) -- the column densities are then bit-identical to
    # the CPU oracle; `value` is measured with the default evaluation (hardware sqrt, fp32 table
    # lerp), which stays within 3e-6 of the fp64 sum (stated tolerance 1e-5).
    exact_ms = None
    if world == 1:
        image = image.clone()          # keep the default-mode frame for the printed statistics
        gh.set_exact_integrals(True)
        trace_once(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            trace_once()
        torch.cuda.synchronize()
        exact_ms = 1e3 * (time.perf_counter() - t1) / 3
        gh.set_exact_integrals(False)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = n_rays * args.steps / elapsed / 1e6
        # Dominant kernel: trace_kernel<cumulative>.  One launch per rank per step.
        # MODE_CUMULATIVE, fast integral; a rank with fewer than 32768 packets of 64 rays runs the
        # class-split instantiation (two or more waves per packet, csrc/trace.hip: the full
        # 1024^2 frame runs two)
        split_kernel = (per + 63) // 64 < 32768
        # (last parameter: the lattice instantiation, for scenes with spheres smaller than the ray
        # spacing -- not this one)
        kernel_name = "trace_kernel<1, %s, true, false>" % ("true" if split_kernel else "false")
        pmc, pmc_note = (None, "PMC passes are collected at N = 1 on the default workload")
        if world == 1 and n == 10_000_000 and args.side == 1024 and args.max_per_leaf == 32:
            pmc, pmc_note = load_pmc(kernel_name)
        kern_s = kern_ms * 1e-3
        alg_per_launch = alg_bytes_total / world
        img = image[:frame_rays]   # frame 0
        # Compulsory bytes: everything the launch must read at least once --
        # the pre-pass records of all particles (16 + 8 B), the cluster boxes (32 B per 64
        # particles) and group boxes (32 B per 4096), this rank's rays (28 B + 4 B of order) --
        # plus 4 B written per ray.  (Axis-aligned packets no longer read the tree's nodes and
        # leaves -- 72 B and 16 B each, counted here up to round 3's flat group passes.)
        compulsory = (24 * n + n // 2 + n // 128 + 36 * (n_rays // world))
        roof = {
            "bound": "valu", "kernel": kernel_name, "unit": "G wave-instr/s",
            "peak": VALU_PEAK_GINST, "achieved": None, "frac": None, "traffic": None,
            "kernel_ms": round(kern_ms, 4), "call_ms": round(call_ms, 4),
            "peak_definition": "1024 SIMDs x 2.4 GHz / 2 cycles per wave64 fp32 VALU instruction "
                               "(MI355X_MICROARCH.md); int/cmp/cvt issue at half and sqrt at a "
                               "quarter of that rate, so a mixed stream cannot reach it",
            "pmc": pmc, "pmc_note": pmc_note,
            # SURVEY.md 8d's figure: every ray's walk counted alone, no packet sharing, no cache
            # credit.  NOT bytes this kernel moves (64 rays share each load; the scene is cache
            # resident) -- kept for reference only.
            "algorithmic_bytes_per_launch_unshared_8d": int(alg_per_launch),
            "algorithmic_GBs_unshared_8d": round(alg_per_launch / kern_s / 1e9, 1),
            "compulsory_bytes_per_launch": int(compulsory),
            "measured_hbm": None, "algorithmic_valu_efficiency": None,
            "per_ray_mean": {"nodes": nodes_v / n_rays, "leaves": leaves_v / n_rays,
                             "spheres_tested": tested / n_rays, "hits": hits / n_rays},
        }
        if pmc:
            insts = pmc["SQ_INSTS_VALU"]
            roof["achieved"] = round(insts / kern_s / 1e9, 1)
            roof["frac"] = round(insts / kern_s / 1e9 / VALU_PEAK_GINST, 4)
            # hits x the minimal instruction count of one ray-sphere evaluation / 64 lanes
            roof["algorithmic_valu_efficiency"] = round(hits * MIN_VALU_PER_HIT / 64.0 / insts, 4)
            if pmc.get("FETCH_SIZE_KB") is not None and pmc.get("WRITE_SIZE_KB") is not None:
                # gfx950: FETCH_SIZE counts 128-B requests as 64 B (guide, HBM section): x2
                traffic = 2.0 * pmc["FETCH_SIZE_KB"] * 1024.0 + pmc["WRITE_SIZE_KB"] * 1024.0
                roof["traffic"] = int(traffic)
                roof["measured_hbm"] = {"GB/s": round(traffic / kern_s / 1e9, 1),
                                        "frac_of_peak": round(traffic / kern_s / 1e9 / HBM_PEAK_GBS, 4),
                                        "peak_GB/s": HBM_PEAK_GBS,
                                        "vs_compulsory": round(traffic / compulsory, 2)}
        out = {
            "metric": "Mrays/s SPH column-density trace, 10^7 particles",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "project_gadget: %d uniform-random SPH particles "
                                   "(h = 48-neighbour rule), %dx%d orthographic -z rays, "
                                   "max_per_leaf %d" % (n, args.side, args.side,
                                                        args.max_per_leaf),
                       "particles": n, "rays": n_rays, "frames": frames,
                       "max_per_leaf": args.max_per_leaf,
                       "call": ("trusted (validation off, caches pinned)" if args.trusted else
                                "cold (caching off)" if args.no_cache else
                                "stateless trace_cumulative_sph(rays, spheres, tree, out); the library's "
                                "validated cache of pre-pass records and ray order is in its steady state"),
                       "sharding": "%d frame(s) of %d rays, contiguous ray shards over %d "
                                   "rank(s), BVH replicated, all_gather of 4 B/ray (two output "
                                   "buffers: step k's gather overlaps step k+1's trace)"
                                   % (frames, frame_rays, world)},
            "roofline": roof,
            "build": dict(phases, n_leaves=tree.n_leaves,
                          total_ms=round(sum(phases.values()), 4)),
            "image": {"mean": float(img.mean()), "max": float(img.max())},
            # every record re-derived per call (grace_trace_set_cache_auto(0)): what the FIRST call
            # on a scene / ray batch costs
            "cold_call_ms": round(cold_ms, 4), "cold_Mrays/s": round(n_rays / world / cold_ms / 1e3, 2),
            # caches pinned, signature validation off (grace_trace_set_cache_validation(0)): the
            # caller's promise instead of the library's check -- round 2's headline configuration
            "trusted_call_ms": round(trusted_ms, 4),
            "value_trusted": round(n_rays / world / trusted_ms / 1e3, 2),
            "bit_exact_integrals": None if exact_ms is None else
                {"ms_per_step": round(exact_ms, 4), "Mrays/s": round(n_rays / exact_ms / 1e3, 2)},
        }
        if world == 1 and not args.no_cpu_baseline:
            gh.trace_release()
            gh.trace_release_rays()
            tree_host = (tree.nodes.cpu().numpy().view(np.float32).reshape(-1, 16),
                         tree.leaves.cpu().numpy(), int(tree.root_index.item()))
            out["cpu_baseline"] = cpu_baseline(spheres.cpu().numpy(), rays.cpu().numpy(), tree_host)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
