"""Round-3 GPU tests: the generic build entry (caller-evaluated boxes, thrust::greater), the
generic extrema, weighted scan in double -- all through the C ABI (ctypes), against the oracle /
numpy."""
import ctypes as C
import math
import os
import subprocess
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "grace-devel_amd", "lib")


def _dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def _sorted_scene(gh, oracle, cuda, n, seed=0):
    s = oracle.random_real4(n, (0, 0, 0, 0.002), (1, 1, 1, 0.03), first=seed * n)
    d = _dev(s, cuda)
    gh.morton_keys30_sort_sph(d, (0, 0, 0), (1, 1, 1))
    deltas = torch.empty(n + 1, dtype=torch.float32, device=cuda)
    gh.euclidean_deltas_sph(d, deltas)
    return d, deltas


@pytest.mark.parametrize("n,mpl", [(3000, 1), (20000, 8), (100000, 32), (5000, 100)])
def test_build_from_caller_boxes_equals_sphere_build(gh, oracle, cuda, n, mpl):
    """grace_albvh_build_ex(GRACE_PRIM_BOX): the generic build_ALBVH(tree, prims, deltas,
    AABBFunc) path (the functor evaluated per primitive outside the library) gives the tree of the
    sphere build -- which is the oracle's (albvh.cuh:986-1021 restated)."""
    d, deltas = _sorted_scene(gh, oracle, cuda, n)
    s = d.cpu().numpy()
    boxes = np.concatenate([s[:, :3] - s[:, 3:4], s[:, :3] + s[:, 3:4]], 1).astype(np.float32)
    t0 = gh.Tree(n, mpl, device=cuda); gh.ALBVH_sph(d, deltas, t0)
    t1 = gh.Tree(n, mpl, device=cuda); gh.build_ALBVH(t1, _dev(boxes, cuda), deltas, gh.PRIM_BOX)
    t2 = gh.Tree(n, mpl, device=cuda); gh.build_ALBVH(t2, d, deltas, gh.PRIM_SPHERE_F4)
    nodes, leaves, root, _ = oracle.albvh(s, deltas.cpu().numpy(), mpl)
    for t in (t0, t1, t2):
        assert np.array_equal(t.leaves.cpu().numpy(), leaves)
        assert np.array_equal(t.nodes.cpu().numpy(), nodes)
        assert int(t.root_index.item()) == root


@pytest.mark.parametrize("dtype", ["f32", "f64", "u32", "u64"])
def test_greater_comparator_is_less_on_flipped_deltas(gh, oracle, cuda, dtype):
    """DeltaComp = thrust::greater (albvh.cuh:1029-1045): delta_comp is only ever applied as
    delta_comp(delta_L, delta_R) (albvh.cuh:129,194,465,607), so building with `greater` on
    order-reversed deltas must reproduce the `less` tree exactly."""
    n, mpl = 40000, 16
    d, deltas = _sorted_scene(gh, oracle, cuda, n, seed=1)
    if dtype == "f32":
        dl, flipped = deltas, -deltas
    elif dtype == "f64":
        dl = deltas.double(); flipped = -dl
    else:
        keys = torch.empty(n, dtype=torch.int32 if dtype == "u32" else torch.int64, device=cuda)
        gh.morton_keys_sph(d, keys, (0, 0, 0), (1, 1, 1))
        dl = torch.empty(n + 1, dtype=keys.dtype, device=cuda)
        gh.XOR_deltas_sph(keys, dl)
        flipped = ~dl
    t_less = gh.Tree(n, mpl, device=cuda); gh.build_ALBVH(t_less, d, dl)
    t_gt = gh.Tree(n, mpl, device=cuda); gh.build_ALBVH(t_gt, d, flipped.contiguous(), delta_comp=gh.COMP_GREATER)
    assert torch.equal(t_less.nodes, t_gt.nodes) and torch.equal(t_less.leaves, t_gt.leaves)
    assert int(t_less.root_index.item()) == int(t_gt.root_index.item())
    with pytest.raises(ValueError):
        gh.build_ALBVH(gh.Tree(n, mpl, device=cuda), d, dl, delta_comp=7)


@pytest.mark.parametrize("dtype,cols", [(np.float32, 4), (np.float32, 3), (np.float64, 4), (np.int32, 4),
                                        (np.float32, 7)])
@pytest.mark.parametrize("n", [1, 63, 1000, 300001])
def test_generic_extrema(gh, cuda, dtype, cols, n):
    rng = np.random.default_rng(n + cols)
    a = (rng.standard_normal((n, cols)) * 1000).astype(dtype)
    d = _dev(a, cuda)
    for first, nc in ((0, 1), (1, 1), (2, 1), (0, 2), (0, 3), (0, min(cols, 4)), (cols - 1, 1)):
        lo, hi = gh.min_max_components(d, nc, first)
        assert np.array_equal(lo, a[:, first:first + nc].min(0)) and np.array_equal(hi, a[:, first:first + nc].max(0))


def test_weighted_segmented_scan_double(gh, cuda):
    """weighted_exclusive_segmented_scan<double> (scan.cuh:43-58): integer-valued data, so the
    sums are exact and must equal the host loop (tests/segmented_scan/segmented_scan.cu's criterion)."""
    rng = np.random.default_rng(4)
    n, n_seg = 200000, 1500
    x = rng.integers(1, 10, n).astype(np.float64)
    w = rng.integers(1, 6, 32).astype(np.float64)
    m = rng.integers(0, 32, n).astype(np.int32)
    offs = np.sort(rng.integers(0, n, n_seg)).astype(np.int32); offs[0] = 0
    out = torch.empty(n, dtype=torch.float64, device=cuda)
    gh.weighted_exclusive_segmented_scan(_dev(x, cuda), _dev(w, cuda), _dev(m, cuda), _dev(offs, cuda), out)
    ref = np.empty(n)
    wx = x * w[m]
    ends = list(offs[1:]) + [n]
    for b, e in zip(offs, ends):
        c = np.cumsum(wx[b:e]); ref[b:e] = c - wx[b:e]
    assert np.array_equal(out.cpu().numpy(), ref)


# ---- contexts, streams ---------------------------------------------------------------------------
def _projection_scene(gh, oracle, cuda, n=150000, side=256, mpl=32):
    s = oracle.random_real4(n, (0, 0, 0, 0.004), (1, 1, 1, 0.02))
    d = _dev(s, cuda)
    lo, hi = gh.min_max_vec4(d)
    lo[3] = hi[3] = 0.0
    tree = gh.Tree(n, mpl, device=cuda)
    gh.build_tree(d, tree, lo[:3], hi[:3])
    rays, _ = gh.orthogonal_rays_z(side, lo, hi, device=cuda)
    return d, tree, rays


def test_trace_after_the_previous_stream_was_destroyed(gh, oracle, cuda):
    """A C-ABI caller may destroy a stream between calls (the workspace fence used to be
    recorded on the PREVIOUS call's stream handle).  Trace on stream A, destroy A, trace on stream
    B, then on the default stream: same bits every time."""
    hip = C.CDLL("libamdhip64.so")
    d, tree, rays = _projection_scene(gh, oracle, cuda)
    ref = torch.empty(len(rays), dtype=torch.float32, device=cuda)
    gh.trace_cumulative_sph(rays, d, tree, ref)
    torch.cuda.synchronize()
    outs = []
    for _ in range(3):
        h = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(h)) == 0
        ext = torch.cuda.ExternalStream(h.value)
        out = torch.empty_like(ref)
        with torch.cuda.stream(ext):
            gh.trace_cumulative_sph(rays, d, tree, out)
            counts = torch.empty(len(rays), dtype=torch.int32, device=cuda)
            gh.trace_hitcounts_sph(rays, d, tree, counts)
        assert hip.hipStreamSynchronize(h) == 0
        assert hip.hipStreamDestroy(h) == 0
        outs.append(out)
    out = torch.empty_like(ref)
    gh.trace_cumulative_sph(rays, d, tree, out)      # default stream, after the last destroy
    gh.trace_status()
    torch.cuda.synchronize()
    for o in outs + [out]:
        assert torch.equal(o, ref)


def test_two_contexts_on_one_gpu_from_two_threads(gh, oracle, cuda):
    """Library state is per context: two host threads, each with a context and
    a stream of its own, trace different shards of one frame concurrently, several times over --
    every result equals the single-context image bit for bit; knobs set in one context do not
    leak into the other."""
    d, tree, rays = _projection_scene(gh, oracle, cuda, n=200000, side=256)
    ref = torch.empty(len(rays), dtype=torch.float32, device=cuda)
    gh.trace_cumulative_sph(rays, d, tree, ref)
    gh.set_exact_integrals(True)
    ref_exact = torch.empty_like(ref)
    gh.trace_cumulative_sph(rays, d, tree, ref_exact)
    gh.set_exact_integrals(False)
    torch.cuda.synchronize()
    assert not torch.equal(ref, ref_exact)
    half = len(rays) // 2
    errors = []

    def worker(k):
        try:
            torch.cuda.set_device(cuda)
            with gh.Context():
                gh.set_exact_integrals(k == 1)        # this context only
                want = (ref_exact if k == 1 else ref)[k * half:(k + 1) * half]
                mine = rays[k * half:(k + 1) * half]
                stream = torch.cuda.Stream(device=cuda)
                with torch.cuda.stream(stream):
                    for it in range(6):
                        out = torch.empty(half, dtype=torch.float32, device=cuda)
                        gh.trace_cumulative_sph(mine, d, tree, out)
                        if it % 2:
                            cnt = torch.empty(half, dtype=torch.int32, device=cuda)
                            gh.trace_hitcounts_sph(mine, d, tree, cnt)
                        stream.synchronize()
                        if not torch.equal(out, want):
                            errors.append((k, it, "differs"))
                    gh.trace_status()
        except Exception as e:      # noqa: BLE001 -- reported to the main thread
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    # the default context still has its own (default) knobs
    again = torch.empty_like(ref)
    gh.trace_cumulative_sph(rays, d, tree, again)
    assert torch.equal(again, ref)


def _build_sharded(tmp_path):
    exe = str(tmp_path / "project_gadget_sharded")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17",
                           "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "tests", "cpp"),
                           os.path.join(ROOT, "tests", "cpp", "project_gadget_sharded.hip"), "-o", exe,
                           "-L" + LIBDIR, "-lgrace_hip", "-L/opt/rocm/lib", "-lrccl", "-pthread",
                           "-Wl,-rpath," + LIBDIR])
    return exe


def test_single_process_sharded_projection(tmp_path, gh, cuda):
    """One process, one thread + one library context per rank (tests/cpp/project_gadget_sharded.hip):
    with one rank the gather is a real ncclAllGather (RCCL, communicator from ncclCommInitAll); with
    three ranks sharing the one GPU of the test box the three contexts trace concurrently and the
    gather is device-to-device copies (RCCL refuses two ranks per device).  Either way the gathered
    image equals the unsharded one, and the ctypes path's, bit for bit."""
    from grace_hip import gadget
    n, side = 120000, 160
    rng = np.random.default_rng(21)
    pos = rng.random((n, 3), dtype=np.float32)
    h = np.full(n, (3 * 48 / (4 * math.pi * n)) ** (1 / 3), np.float32) * (0.8 + 0.4 * rng.random(n, dtype=np.float32))
    snap = str(tmp_path / "snap")
    gadget.write_gadget(snap, pos, h)
    exe = _build_sharded(tmp_path)
    s = torch.from_numpy(gadget.read_gadget(snap)).to(cuda)
    lo, hi = gh.min_max_vec4(s)
    lo[3] = hi[3] = 0.0
    tree = gh.Tree(n, 32, device=cuda)
    gh.build_tree(s, tree, lo[:3], hi[:3])
    rays, _ = gh.orthogonal_rays_z(side, lo, hi, device=cuda)
    img = torch.empty(len(rays), dtype=torch.float32, device=cuda)
    gh.trace_cumulative_sph(rays, s, tree, img)
    for extra in (["1"], ["3", "share"]):
        out = str(tmp_path / ("img_%s.f32" % extra[0]))
        r = subprocess.run([exe, str(side * side // 32), "32", snap, out] + extra, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0 and "PASSED" in r.stdout, r.stdout + r.stderr
        assert np.array_equal(np.fromfile(out, np.float32).view(np.uint32), img.cpu().numpy().view(np.uint32))


# ---- cached trace records are validated, never trusted on pointer equality ----------------------
def _fresh_image(gh, rays, d, tree, cuda):
    """The image as a context without any cache derives it."""
    out = torch.empty(len(rays), dtype=torch.float32, device=cuda)
    with gh.Context():
        gh.trace_cumulative_sph(rays, d, tree, out)
        torch.cuda.synchronize()
    return out


def test_scene_changed_in_place_under_the_cache(gh, oracle, cuda):
    """VERDICT r2 / ADVICE r2: the cached scene records were matched by (pointer, size) only.  Trace a
    scene three times (the third call runs on cached records), then overwrite spheres AND tree IN
    PLACE with another scene of the same size and trace again: the result must be the new scene's,
    bit for bit, and the brute-force hit counts must agree."""
    n, side, mpl = 60000, 128, 16
    s1 = oracle.random_real4(n, (0, 0, 0, 0.004), (1, 1, 1, 0.03))
    d = _dev(s1, cuda)
    tree = gh.Tree(n, mpl, device=cuda)
    nodes0, leaves0 = tree.nodes, tree.leaves          # full-capacity buffers: rebuilt in place below
    gh.build_tree(d, tree, (0, 0, 0), (1, 1, 1))
    rays, _ = gh.orthogonal_rays_z(side, (0, 0, 0, 0), (1, 1, 1, 0.05), device=cuda)
    outs = []
    for _ in range(3):
        o = torch.empty(len(rays), dtype=torch.float32, device=cuda)
        gh.trace_cumulative_sph(rays, d, tree, o)
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # the same device arrays, new contents: every radius grows by 70 % (same centres, hence the same
    # Morton order, deltas and tree topology -- pointers AND sizes of all three arrays stay the same)
    d[:, 3] *= 1.7
    tree2 = gh.Tree.__new__(gh.Tree)
    tree2.max_per_leaf = mpl; tree2.nodes = nodes0; tree2.leaves = leaves0; tree2.root_index = tree.root_index
    gh.build_tree(d, tree2, (0, 0, 0), (1, 1, 1))
    same_arrays = tree2.nodes.data_ptr() == tree.nodes.data_ptr() and tree2.n_leaves == tree.n_leaves
    got = torch.empty(len(rays), dtype=torch.float32, device=cuda)
    gh.trace_cumulative_sph(rays, d, tree2, got)
    counts = torch.empty(len(rays), dtype=torch.int32, device=cuda)
    gh.trace_hitcounts_sph(rays, d, tree2, counts)
    gh.trace_status()
    want = _fresh_image(gh, rays, d, tree2, cuda)
    assert same_arrays, "the test must present the cache with identical pointers and sizes"
    assert torch.equal(got, want), "stale cached scene records were used"
    assert not torch.equal(got, outs[0])
    sub = np.arange(0, len(rays), 37)
    assert np.array_equal(counts.cpu().numpy()[sub], oracle.brute_hitcounts(rays.cpu().numpy()[sub], d.cpu().numpy()))
    # and once more: the re-derived records are now the cached ones
    again = torch.empty_like(got)
    gh.trace_cumulative_sph(rays, d, tree2, again)
    assert torch.equal(again, want)


def test_rays_changed_in_place_under_the_cache(gh, oracle, cuda):
    """The cached coherence order of a ray batch is validated too: new rays at the same address
    (another view direction, even another ray count's worth of geometry) are ordered afresh; the
    per-ray results never depended on the order anyway, so also compare with the caller's order."""
    d, tree, rays = _projection_scene(gh, oracle, cuda, n=100000, side=128)
    for _ in range(3):
        o = torch.empty(len(rays), dtype=torch.float32, device=cuda)
        gh.trace_cumulative_sph(rays, d, tree, o)
    r = rays.cpu().numpy().copy()
    r2 = r.copy()
    r2[:, 0], r2[:, 1], r2[:, 2] = 1.0, 0.0, 0.0                 # +x rays from the x = -1 plane
    r2[:, 3], r2[:, 4], r2[:, 5] = -1.0, r[:, 3], r[:, 4]
    r2[:, 6] = 3.0
    rays.copy_(torch.from_numpy(r2).to(cuda))                    # same address, new batch
    got = torch.empty(len(rays), dtype=torch.float32, device=cuda)
    gh.trace_cumulative_sph(rays, d, tree, got)
    gh.set_ray_reorder(False)
    plain = torch.empty_like(got)
    gh.trace_cumulative_sph(rays, d, tree, plain)
    gh.set_ray_reorder(True)
    gh.trace_status()
    assert torch.equal(got, plain)
    want = _fresh_image(gh, rays, d, tree, cuda)
    assert torch.equal(got, want)
    assert float(got.sum()) > 0


def test_freed_and_reallocated_at_the_same_address(gh, oracle, cuda):
    """An allocator that hands out the address of a freed array again (torch's caching allocator
    does) must not resurrect the freed array's cached records."""
    n, side, mpl = 40000, 96, 8
    rays, _ = gh.orthogonal_rays_z(side, (0, 0, 0, 0), (1, 1, 1, 0.05), device=cuda)
    imgs, ptrs = [], []
    for k in range(3):
        s = oracle.random_real4(n, (0, 0, 0, 0.004), (1, 1, 1, 0.03), first=k * n)
        d = _dev(s, cuda)
        tree = gh.Tree(n, mpl, device=cuda)
        gh.build_tree(d, tree, (0, 0, 0), (1, 1, 1))
        o = torch.empty(len(rays), dtype=torch.float32, device=cuda)
        for _ in range(3):
            gh.trace_cumulative_sph(rays, d, tree, o)
        imgs.append(o.clone()); ptrs.append(d.data_ptr())
        want = _fresh_image(gh, rays, d, tree, cuda)
        assert torch.equal(o, want)
        del d, tree, s
    assert not torch.equal(imgs[0], imgs[1])


def test_trusted_mode_and_pinned_caches(gh, oracle, cuda):
    """grace_trace_set_cache_validation(0): cached records are used on the caller's promise (the
    round-2 prepared scene / prepared rays); bits equal the validated and the uncached results, and
    the library's own writers drop a cache when they write to one of its arrays."""
    d, tree, rays = _projection_scene(gh, oracle, cuda, n=80000, side=128)
    ref = _fresh_image(gh, rays, d, tree, cuda)
    try:
        gh.set_cache_validation(False)
        gh.trace_prepare(d, tree)
        gh.trace_prepare_rays(rays)
        for _ in range(2):
            o = torch.empty_like(ref)
            gh.trace_cumulative_sph(rays, d, tree, o)
            assert torch.equal(o, ref)
        # the library's sort writes the spheres: the trusted cache must be dropped, not reused
        keys = torch.randint(0, 1 << 20, (len(d),), dtype=torch.int32, device=cuda)
        gh.sort_by_key(keys, d)
        tree2 = gh.Tree(len(d), 32, device=cuda)
        gh.build_tree(d, tree2, (0, 0, 0), (1, 1, 1))
        o = torch.empty_like(ref)
        gh.trace_cumulative_sph(rays, d, tree2, o)
        gh.trace_status()
        # (the same spheres in Morton order again -- up to the order of spheres with equal keys, i.e.
        # up to the last bits of the sums: compare with an uncached trace of the new arrangement)
        assert torch.equal(o, _fresh_image(gh, rays, d, tree2, cuda))
        assert torch.allclose(o, ref, rtol=1e-5)
    finally:
        gh.set_cache_validation(True)
        gh.trace_release(); gh.trace_release_rays()
    gh.set_cache_auto(False)
    try:
        want = _fresh_image(gh, rays, d, tree2, cuda)
        for _ in range(3):
            o = torch.empty_like(ref)
            gh.trace_cumulative_sph(rays, d, tree2, o)
            assert torch.equal(o, want)
    finally:
        gh.set_cache_auto(True)


# ---- bucket sort (inputs >= 2^17 elements): one MSD pass with the payload + per-bucket LDS sort;
# ---- a bucket beyond a workgroup's capacity turns on the gated index sort instead (csrc/sort.hip)

def _check_sort(gh, cuda, keys, vals, begin, end, want_perm=True):
    kd = torch.from_numpy(keys.view(np.int32 if keys.dtype == np.uint32 else np.int64)).to(cuda)
    vd = torch.from_numpy(vals).to(cuda) if vals is not None else None
    perm = gh.sort_by_key(kd, vd, begin, end, want_perm=want_perm)
    width = np.uint64(end - begin)
    digit = (keys.astype(np.uint64) >> np.uint64(begin)) & ((np.uint64(1) << width) - np.uint64(1))
    order = np.argsort(digit, kind="stable")
    assert np.array_equal(kd.cpu().numpy().view(keys.dtype), keys[order])
    if vals is not None:
        assert np.array_equal(vd.cpu().numpy(), vals[order])
    if want_perm:
        assert np.array_equal(perm.cpu().numpy().view(np.uint32), order.astype(np.uint32))


@pytest.mark.parametrize("words", [0, 1, 2, 3, 4, 7, 8, 9])
@pytest.mark.parametrize("n", [262144, 262145, 300001, 1 << 20])
def test_bucket_sort_uniform_keys(gh, cuda, n, words):
    rng = np.random.default_rng(n + words)
    keys = rng.integers(0, 1 << 30, n, dtype=np.uint32)
    keys[: n // 50] &= 0x3FF00000          # duplicates inside buckets: stability matters
    vals = rng.integers(0, 1 << 31, (n, words), dtype=np.int32) if words else None
    _check_sort(gh, cuda, keys, vals, 0, 30, want_perm=(words in (0, 4)))


@pytest.mark.parametrize("case", ["one_value", "few_values", "half_in_one_bucket", "sorted", "reversed"])
def test_bucket_sort_overflowing_buckets(gh, cuda, case):
    n = 700001
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 1 << 30, n, dtype=np.uint32)
    if case == "one_value":
        keys[:] = 0x2AAAAAAA
    elif case == "few_values":
        keys = rng.integers(0, 37, n, dtype=np.uint32) << np.uint32(13)
    elif case == "half_in_one_bucket":
        keys[: n // 2] = (keys[: n // 2] & np.uint32(0x3FFFF)) | np.uint32(0x155 << 18)
    elif case == "sorted":
        keys = np.sort(keys)
    elif case == "reversed":
        keys = np.sort(keys)[::-1].copy()
    vals = rng.integers(0, 1 << 31, (n, 4), dtype=np.int32)
    _check_sort(gh, cuda, keys, vals, 0, 30)
    # a second, uniform call right after (same workspace, flags recomputed)
    keys2 = rng.integers(0, 1 << 30, n, dtype=np.uint32)
    _check_sort(gh, cuda, keys2, vals, 0, 30)


@pytest.mark.parametrize("begin,end", [(0, 32), (3, 27), (10, 30), (0, 17), (12, 20), (0, 8)])
def test_bucket_sort_bit_ranges(gh, cuda, begin, end):
    n = 400003
    rng = np.random.default_rng(begin * 100 + end)
    keys = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    vals = rng.integers(0, 1 << 31, (n, 2), dtype=np.int32)
    _check_sort(gh, cuda, keys, vals, begin, end)


@pytest.mark.parametrize("begin,end", [(0, 63), (0, 30), (20, 52), (40, 64)])
def test_bucket_sort_u64(gh, cuda, begin, end):
    n = 300007
    rng = np.random.default_rng(end)
    keys = rng.integers(0, 1 << 63, n, dtype=np.uint64)
    if end == 64:
        keys |= rng.integers(0, 2, n, dtype=np.uint64) << np.uint64(63)
    keys[::11] = keys[5]
    vals = rng.integers(0, 1 << 31, (n, 4), dtype=np.int32)
    _check_sort(gh, cuda, keys, vals, begin, end)


def test_bucket_sort_full_size_properties(gh, cuda):
    """10^7 keys (BASELINE config 4's build): sortedness, stability through the carried index,
    payload follows its key."""
    n = 10_000_000
    g = torch.Generator(device=cuda); g.manual_seed(3)
    keys = torch.randint(0, 1 << 30, (n,), dtype=torch.int32, device=cuda, generator=g)
    keys[: n // 4] &= 0x3FFFFF00
    vals = torch.empty((n, 4), dtype=torch.int32, device=cuda)
    vals[:, 0] = keys; vals[:, 1] = torch.arange(n, dtype=torch.int32, device=cuda)
    vals[:, 2] = 7; vals[:, 3] = -keys
    gh.sort_by_key(keys, vals, 0, 30)
    assert bool((keys[1:] >= keys[:-1]).all())
    assert torch.equal(vals[:, 0], keys) and torch.equal(vals[:, 3], -keys)
    same = keys[1:] == keys[:-1]
    assert bool((vals[1:, 1][same] > vals[:-1, 1][same]).all())       # equal keys keep input order
    assert int(vals[:, 1].to(torch.int64).sum()) == n * (n - 1) // 2


def test_bucket_sort_two_contexts_two_threads_and_context_teardown(gh, cuda):
    """The bucket sort's fallback runs on the context's side stream: two threads with a context
    and a stream each sort concurrently (one of them with overflowing buckets, so that its gated
    index sort really runs beside the other's bucket kernels); leaving the `with` destroys the
    contexts, side streams included."""
    n = 500_003
    rng = np.random.default_rng(11)
    errors = []

    def worker(k):
        try:
            torch.cuda.set_device(cuda)
            r = np.random.default_rng(100 + k)
            with gh.Context():
                stream = torch.cuda.Stream(device=cuda)
                with torch.cuda.stream(stream):
                    for it in range(5):
                        keys = r.integers(0, 1 << 30, n, dtype=np.uint32)
                        if k == 1 and it % 2 == 0:
                            keys[: n // 2] = (keys[: n // 2] & np.uint32(0xFFFF)) | np.uint32(0x2AA << 20)
                        vals = r.integers(0, 1 << 31, (n, 4), dtype=np.int32)
                        kd = torch.from_numpy(keys.view(np.int32)).to(cuda, non_blocking=False)
                        vd = torch.from_numpy(vals).to(cuda)
                        gh.sort_by_key(kd, vd, 0, 30)
                        stream.synchronize()
                        order = np.argsort(keys, kind="stable")
                        if not (np.array_equal(kd.cpu().numpy().view(np.uint32), keys[order])
                                and np.array_equal(vd.cpu().numpy(), vals[order])):
                            errors.append((k, it, "differs"))
        except Exception as e:      # noqa: BLE001 -- reported to the main thread
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    # and the default context still sorts
    keys = rng.integers(0, 1 << 30, n, dtype=np.uint32)
    _check_sort(gh, cuda, keys, None, 0, 30)


# ---- flat group passes (csrc/trace_kernel.hpp): hit counts and column densities test the boxes of
# ---- groups of 4096 Morton-consecutive primitives instead of walking the tree

@pytest.mark.parametrize("n", [33, 4095, 4096, 4097, 8191, 12289, 70001])
@pytest.mark.parametrize("kind", ["axis+", "axis-", "pencil", "general"])
def test_group_passes_equal_brute_force(gh, oracle, cuda, n, kind):
    """Scene sizes around the group boundaries (one group, exactly one, one primitive into the
    next, a short last group, many groups), every packet kind that takes the group passes, rays
    that stop short of / start inside the box (the extent test along the axis), and every split of
    the packets: hit counts equal the brute-force loop, column densities the oracle's sums."""
    rng = np.random.default_rng(n)
    s = oracle.random_real4(n, (0, 0, 0, 0.01), (1, 1, 1, 0.06))
    d = torch.from_numpy(s).to(cuda)
    tree = gh.Tree(n, 32 if n > 64 else 2, device=cuda)
    gh.build_tree(d, tree, (0, 0, 0), (1, 1, 1))
    ss = d.cpu().numpy()                      # sorted in place by the build
    side = 40
    u, v = np.meshgrid((np.arange(side) + 0.5) / side, (np.arange(side) + 0.5) / side)
    rays = np.zeros((side * side, 7), np.float32)
    if kind.startswith("axis"):
        sense = 1.0 if kind == "axis+" else -1.0
        rays[:, 2] = sense
        rays[:, 3] = u.ravel(); rays[:, 4] = v.ravel()
        start = rng.uniform(-0.2, 0.7, len(rays)).astype(np.float32)
        rays[:, 5] = start if sense > 0 else (1.0 - start)
        rays[:, 6] = rng.uniform(0.05, 1.5, len(rays)).astype(np.float32)
    else:
        dirs = rng.standard_normal((len(rays), 3)); dirs /= np.linalg.norm(dirs, axis=1)[:, None]
        rays[:, :3] = dirs.astype(np.float32)
        if kind == "pencil":
            rays[:, 3:6] = np.float32(0.5)
        else:
            rays[:, 3:6] = rng.uniform(0.2, 0.8, (len(rays), 3)).astype(np.float32)
        rays[:, 6] = rng.uniform(0.1, 1.2, len(rays)).astype(np.float32)
    dr = torch.from_numpy(rays).to(cuda)
    want_c = oracle.brute_hitcounts(rays, ss)
    ref32, ref64 = oracle.brute_cumulative(rays, ss)
    max_term = float(oracle.kernel_table()[0]) / 0.01 ** 2
    from conftest import check_column_densities
    try:
        for split in (-1, 1, 4, 8):
            gh.set_packet_split(split)
            for exact in (False, True):
                gh.set_exact_integrals(exact)
                hc = torch.empty(len(rays), dtype=torch.int32, device=cuda)
                cu = torch.empty(len(rays), dtype=torch.float32, device=cuda)
                gh.trace_hitcounts_sph(dr, d, tree, hc)
                gh.trace_cumulative_sph(dr, d, tree, cu)
                gh.trace_status()
                assert np.array_equal(hc.cpu().numpy(), want_c), (split, exact)
                check_column_densities(cu.cpu().numpy(), ref32, ref64, "exact" if exact else "fast",
                                       max_term=max_term)
    finally:
        gh.set_packet_split(-1); gh.set_exact_integrals(False)


def test_group_passes_fall_back_to_the_walk_for_wide_packets(gh, oracle, cuda):
    """A packet that keeps more than 256 groups (here: two packets' worth of isotropic rays from
    one origin, and from two origins, through 1.5 M small spheres = 367 groups) walks the tree
    instead: the same counts and sums as the brute-force loop either way."""
    n = 1_500_000
    s = oracle.random_real4(n, (0, 0, 0, 0.002), (1, 1, 1, 0.012))
    d = torch.from_numpy(s).to(cuda)
    tree = gh.Tree(n, 32, device=cuda)
    gh.build_tree(d, tree, (0, 0, 0), (1, 1, 1))
    ss = d.cpu().numpy()
    rng = np.random.default_rng(3)
    from conftest import check_column_densities
    for origins in ([(0.5, 0.5, 0.5)], [(0.3, 0.3, 0.3), (0.7, 0.6, 0.5)]):
        R = 128
        dirs = rng.standard_normal((R, 3)); dirs /= np.linalg.norm(dirs, axis=1)[:, None]
        rays = np.zeros((R, 7), np.float32)
        rays[:, :3] = dirs.astype(np.float32)
        rays[:, 3:6] = np.array(origins, np.float32)[np.arange(R) % len(origins)]
        rays[:, 6] = 2.0
        dr = torch.from_numpy(rays).to(cuda)
        hc = torch.empty(R, dtype=torch.int32, device=cuda)
        cu = torch.empty(R, dtype=torch.float32, device=cuda)
        gh.trace_hitcounts_sph(dr, d, tree, hc)
        gh.trace_cumulative_sph(dr, d, tree, cu)
        gh.trace_status()
        assert np.array_equal(hc.cpu().numpy(), oracle.brute_hitcounts(rays, ss))
        ref32, ref64 = oracle.brute_cumulative(rays, ss)
        check_column_densities(cu.cpu().numpy(), ref32, ref64, "fast")


@pytest.mark.parametrize("repeat", [1, 3, 8, 9, 40])
@pytest.mark.parametrize("words", [0, 4])
def test_bucket_sort_long_keys_tie_runs(gh, cuda, repeat, words):
    """63-bit keys: the bucket kernel sorts the top 24 bits below the bucket digit by LDS passes and
    settles the rest inside runs of records that agree on everything above -- `repeat` records per
    36-bit prefix, differing (or not) in the 27 bits below: single records, short runs (insertion
    sort), runs beyond 8 (the bucket is re-sorted over all its bits)."""
    n = 400_000 // repeat * repeat
    rng = np.random.default_rng(repeat + 10 * words)
    prefixes = rng.integers(0, 1 << 36, n // repeat, dtype=np.uint64)
    keys = (np.repeat(prefixes, repeat) << np.uint64(27)) | rng.integers(0, 1 << 27, n, dtype=np.uint64)
    keys[::5] &= ~np.uint64(0x7FFFFFF)                       # some fully equal keys too: stability
    keys = keys[rng.permutation(n)]
    vals = rng.integers(0, 1 << 31, (n, words), dtype=np.int32) if words else None
    _check_sort(gh, cuda, keys, vals, 0, 63, want_perm=(words == 0))


def test_bucket_sort_overflow_hint(gh, cuda):
    """With the hint on, a context whose large sort overflowed goes straight to the index sort for the
    next large sorts and looks again at the 8th; results are the same on either path."""
    n = 400_000
    rng = np.random.default_rng(9)
    uniform = rng.integers(0, 1 << 30, n, dtype=np.uint32)
    clustered = uniform.copy()
    clustered[: n // 2] = (clustered[: n // 2] & np.uint32(0xFFFF)) | np.uint32(0x155 << 20)
    vals = rng.integers(0, 1 << 31, (n, 4), dtype=np.int32)
    gh.set_sort_overflow_hint(True)
    try:
        with gh.Context():
            _check_sort(gh, cuda, uniform, vals, 0, 30)
            _check_sort(gh, cuda, clustered, vals, 0, 30)        # overflows: hint set
            torch.cuda.synchronize()
            for _ in range(10):                                   # skipped 7 times, retried, cleared
                _check_sort(gh, cuda, uniform, vals, 0, 30)
                torch.cuda.synchronize()
            _check_sort(gh, cuda, clustered, vals, 0, 30)
    finally:
        gh.set_sort_overflow_hint(False)
