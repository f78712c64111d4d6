#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the SPH column-density trace (trace_cumulative_sph
semantics) on the project_gadget workload -- 10^7 particles, 1024^2 orthographic rays
(BASELINE.json configs[3], the configuration the metric is quoted on; it fits one GPU).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed / RCCL when N > 1): every rank builds the same BVH
from the same seeded particles (the tree is replicated), traces its own contiguous shard of
the job's ray batch, and the per-ray integrals are all-gathered over RCCL (the only
collective; 4 B/ray).  A "step" = one trace of the whole ray batch (+ the gather when N > 1)
with particles, tree and rays resident in HBM.

  --scaling weak (default): the job is N frames of the 1024^2 image -- frame 0 is the
      pixel-centre grid of configs[3], frame r > 0 the same grid shifted by a fixed sub-pixel
      offset (an N-sample supersampled projection) -- so every rank traces 1024^2 rays of the
      same statistical character whatever N is; a rank's contiguous shard is its frame.
  --scaling strong: the job is the single 1024^2 frame, cut into N contiguous shards.

Rank 0 prints ONE JSON line.  The oracle is used only by the cpu_baseline leg (rank 0,
N = 1) as the thing timed on the host cores, never in the GPU path.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "grace-devel_amd"))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def make_particles(n, device, seed=42):
    """Synthetic SPH snapshot: n particles uniform in the unit box, smoothing length from
    the ~48-neighbour rule h = (3*48 / (4 pi n))^(1/3)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    s = torch.empty((n, 4), dtype=torch.float32, device=device)
    s[:, :3] = torch.rand((n, 3), generator=g, device=device, dtype=torch.float32)
    s[:, 3] = float((3.0 * 48.0 / (4.0 * math.pi * n)) ** (1.0 / 3.0))
    return s


def cpu_baseline(spheres_host, rays_host, seconds_target=15.0):
    """The reference's host-side tree_traversal path (OpenMP brute-force sphere_hit over
    every (ray, sphere) pair, tests/tree_traversal/tree_traversal.cu:65-79) with the
    column-density accumulation, timed on this box's cores on a bounded ray sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    O.build()
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    n_rays = len(rays_host)
    # Probe, then size the sample for ~seconds_target of work.
    probe = rays_host[np.linspace(0, n_rays - 1, 4 * cores).astype(np.int64)]
    t0 = time.perf_counter(); O.brute_cumulative(probe, spheres_host); t1 = time.perf_counter()
    per_ray = max((t1 - t0) / len(probe), 1e-9)
    n_sample = int(min(n_rays, max(4 * cores, seconds_target / per_ray)))
    n_sample = max(cores, (n_sample // cores) * cores)
    sample = rays_host[np.linspace(0, n_rays - 1, n_sample).astype(np.int64)]
    t0 = time.perf_counter(); O.brute_cumulative(sample, spheres_host); t1 = time.perf_counter()
    rate = n_sample / (t1 - t0)
    # BASELINE.md's second host path: the tests/morton_key loop (morton_key(float,float,float),
    # generic/morton.h:32-42) over the same particles, one thread (the loop is serial there too).
    nk = min(len(spheres_host), 10_000_000)
    bot = spheres_host[:nk, :3].min(axis=0); top = spheres_host[:nk, :3].max(axis=0)
    t2 = time.perf_counter(); O.morton_keys30(spheres_host[:nk], bot, top); t3 = time.perf_counter()
    return {"value": rate / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "morton_key_Mkeys_per_s_1thread": nk / (t3 - t2) / 1e6,
            "sample": "%d of %d rays (evenly spaced over the image) x %d spheres, brute-force "
                      "sphere_hit + kernel-integral accumulation, %.1f s wall"
                      % (n_sample, n_rays, len(spheres_host), t1 - t0),
            "pair_tests_per_s": rate * len(spheres_host)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--particles", type=int, default=10_000_000)
    ap.add_argument("--side", type=int, default=1024)
    ap.add_argument("--max-per-leaf", type=int, default=32)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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

    def step():
        gh.trace_cumulative_sph(my_rays, spheres, tree, my_out[: r1 - r0])
        return sharding.gather_results(my_out, n_rays, world, dist)

    for _ in range(args.warmup):
        image = step()
    # HIP events directly around the traversal kernel on its stream (inside the library);
    # read back after each step's launch -- the wait falls on the kernel that is timed anyway.
    gh.enable_kernel_timing(True)
    trace_ms = []
    kern_ev = [(ev(), ev()) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        kern_ev[k][0].record()
        gh.trace_cumulative_sph(my_rays, spheres, tree, my_out[: r1 - r0])
        kern_ev[k][1].record()
        image = sharding.gather_results(my_out, n_rays, world, dist)
        trace_ms.append(gh.last_kernel_ms())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    call_ms = sum(a.elapsed_time(b) for a, b in kern_ev) / args.steps   # pre-passes + kernel
    kern_ms = sum(trace_ms) / len(trace_ms)                             # trace_kernel alone
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
        gh.trace_cumulative_sph(my_rays, spheres, tree, my_out[: r1 - r0])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            gh.trace_cumulative_sph(my_rays, spheres, tree, my_out[: r1 - r0])
        torch.cuda.synchronize()
        exact_ms = 1e3 * (time.perf_counter() - t1) / 3
        gh.set_exact_integrals(False)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = n_rays * args.steps / elapsed / 1e6
        # Dominant kernel: trace_kernel<cumulative>.  One launch per rank per step; its
        # algorithmic bytes are this job's total divided over the ranks.
        alg_per_launch = alg_bytes_total / world
        achieved = alg_per_launch / (kern_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tfile) and world == 1 and n == 10_000_000 and args.side == 1024:
            try:
                traffic = json.load(open(tfile)).get("trace_cumulative_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        img = image[:frame_rays]   # frame 0
        # Compulsory bytes: everything the launch must read at least once --
        # the pre-pass records of all particles (16 + 8 B), every node (64 B + 8 B span) and
        # leaf (16 B), this rank's rays (28 B) -- plus 4 B written per ray.
        compulsory = (24 * n + 72 * (tree.n_leaves - 1) + 16 * tree.n_leaves
                      + 32 * (n_rays // world))
        measured = None
        if traffic:
            measured = {"GB/s": round(traffic / (kern_ms * 1e-3) / 1e9, 1),
                        "frac_of_peak": round(traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
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
                       "sharding": "%d frame(s) of %d rays, contiguous ray shards over %d "
                                   "rank(s), BVH replicated, all_gather of 4 B/ray"
                                   % (frames, frame_rays, world)},
            "roofline": {"bound": "hbm", "kernel": "trace_kernel<cumulative>",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel_ms": round(kern_ms, 4), "call_ms": round(call_ms, 4),
                         "algorithmic_bytes_per_launch": int(alg_per_launch),
                         "compulsory_bytes_per_launch": int(compulsory),
                         "measured_hbm": measured,
                         "per_ray_mean": {"nodes": nodes_v / n_rays, "leaves": leaves_v / n_rays,
                                          "spheres_tested": tested / n_rays,
                                          "hits": hits / n_rays}},
            "build": dict(phases, n_leaves=tree.n_leaves,
                          total_ms=round(sum(phases.values()), 4)),
            "image": {"mean": float(img.mean()), "max": float(img.max())},
            "bit_exact_integrals": None if exact_ms is None else
                {"ms_per_step": round(exact_ms, 4), "Mrays/s": round(n_rays / exact_ms / 1e3, 2)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spheres.cpu().numpy(), rays.cpu().numpy())
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
