"""GPU parity: every stage of the hot path through the C ABI against the CPU oracle.

Bit-exact for keys, sort order, deltas, leaves, nodes, hit counts; the cumulative integral
is compared bit-for-bit with the oracle's fp32 sum taken in the same (ascending primitive)
order, and to 1e-5 relative against its fp64 accumulation (BASELINE.md tolerance).
"""
import numpy as np
import pytest
from conftest import check_column_densities
import torch

pytestmark = pytest.mark.gpu


def _spheres(O, n, lo=(0, 0, 0, 0), hi=(1, 1, 1, 0.1)):
    # tests/hitcounts/hitcounts.cu:47-58 : centres U[0,1)^3, radii U[0,0.1)
    return O.random_real4(n, lo, hi)


def _dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


@pytest.mark.parametrize("n", [1, 2, 777, 4096, 100000])
def test_morton_keys30(gh, oracle, cuda, n):
    s = _spheres(oracle, n)
    bot, top = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
    keys = torch.empty(n, dtype=torch.int32, device=cuda)
    gh.morton_keys_sph(_dev(s, cuda), keys, bot, top)
    ref = oracle.morton_keys30(s, bot, top)
    assert np.array_equal(keys.cpu().numpy().view(np.uint32), ref)


def test_morton_keys63_and_bounds(gh, oracle, cuda):
    n = 50000
    s = _spheres(oracle, n, (-3, 2, 10, 0), (5, 9, 11, 1))
    d = _dev(s, cuda)
    bot, top = gh.centroid_bounds(d)
    rb, rt = oracle.centroid_bounds(s)
    assert np.array_equal(bot, rb) and np.array_equal(top, rt)
    keys = torch.empty(n, dtype=torch.int64, device=cuda)
    gh.morton_keys_sph(d, keys, bot, top)
    assert np.array_equal(keys.cpu().numpy().view(np.uint64), oracle.morton_keys63(s, bot, top))
    gh.morton_keys_sph(d, keys, bot, top, double_bounds=True)
    assert np.array_equal(keys.cpu().numpy().view(np.uint64),
                          oracle.morton_keys63(s, bot, top, double_bounds=True))
    # bounds-free overload == explicit bounds (build_sph.cuh:19-25)
    k32 = torch.empty(n, dtype=torch.int32, device=cuda)
    gh.morton_keys_sph(d, k32)
    assert np.array_equal(k32.cpu().numpy().view(np.uint32), oracle.morton_keys30(s, rb, rt))


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 4095, 4096, 4097, 123457])
def test_sort_stable_u32(gh, cuda, n):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 1 << 30, n, dtype=np.uint32)
    keys[: n // 3] &= 0xFF  # many duplicates: stability matters
    vals = rng.standard_normal((n, 4)).astype(np.float32)
    dk = _dev(keys.view(np.int32), cuda); dv = _dev(vals, cuda)
    perm = gh.sort_by_key(dk, dv, 0, 30, want_perm=True)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(dk.cpu().numpy().view(np.uint32), keys[order])
    assert np.array_equal(perm.cpu().numpy().view(np.uint32), order.astype(np.uint32))
    assert np.array_equal(dv.cpu().numpy(), vals[order])


@pytest.mark.parametrize("vbytes", [4, 28, 32, 36])
def test_sort_u64_payloads(gh, cuda, vbytes):
    n = 20011
    rng = np.random.default_rng(vbytes)
    keys = rng.integers(0, 1 << 63, n, dtype=np.uint64)
    keys[::7] = keys[0]
    vals = rng.integers(0, 1 << 31, (n, vbytes // 4), dtype=np.int32)
    dk = _dev(keys.view(np.int64), cuda); dv = _dev(vals, cuda)
    gh.sort_by_key(dk, dv, 0, 63)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(dk.cpu().numpy().view(np.uint64), keys[order])
    assert np.array_equal(dv.cpu().numpy(), vals[order])


def test_deltas(gh, oracle, cuda):
    n = 30001
    s = _spheres(oracle, n)
    d = _dev(s, cuda)
    out = torch.empty(n + 1, dtype=torch.float32, device=cuda)
    gh.euclidean_deltas_sph(d, out)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), oracle.deltas_euclid(s).view(np.uint32))
    gh.surface_area_deltas_sph(d, out)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), oracle.deltas_area(s).view(np.uint32))
    keys = oracle.morton_keys30(s, (0, 0, 0), (1, 1, 1))
    keys.sort()
    dk = _dev(keys.view(np.int32), cuda)
    xo = torch.empty(n + 1, dtype=torch.int32, device=cuda)
    gh.XOR_deltas_sph(dk, xo)
    assert np.array_equal(xo.cpu().numpy().view(np.uint32), oracle.deltas_xor(keys))


def _build_both(gh, oracle, cuda, s, mpl, low=(0, 0, 0), high=(1, 1, 1)):
    n = len(s)
    d = _dev(s, cuda)
    tree = gh.Tree(n, mpl, device=cuda)
    gh.build_tree(d, tree, low, high)
    keys = oracle.morton_keys30(s, low, high)
    _, ss, _ = oracle.sort_by_key(keys, s)
    ss = np.ascontiguousarray(ss)
    nodes, leaves, root, _ = oracle.albvh(ss, oracle.deltas_euclid(ss), mpl)
    return d, tree, ss, nodes, leaves, root


@pytest.mark.parametrize("n,mpl", [(2, 1), (3, 1), (50, 1), (4096, 1), (4096, 8), (4096, 32),
                                   (33, 32), (100000, 32), (100000, 5)])
def test_albvh_tree_identical(gh, oracle, cuda, n, mpl):
    s = _spheres(oracle, n)
    d, tree, ss, nodes, leaves, root = _build_both(gh, oracle, cuda, s, mpl)
    assert np.array_equal(d.cpu().numpy(), ss)                      # sorted primitives
    assert np.array_equal(tree.leaves.cpu().numpy(), leaves)        # leaves bit-exact
    assert int(tree.root_index.item()) == root
    assert np.array_equal(tree.nodes.cpu().numpy(), nodes)          # children, ranges, AABBs


def test_albvh_xor_deltas_and_duplicates(gh, oracle, cuda):
    # coincident primitives => equal keys and zero deltas: exercises the tie rule
    n = 5000
    s = _spheres(oracle, n)
    s[100:400] = s[100]
    s[1000:1010, :3] = s[1000, :3]
    keys = oracle.morton_keys30(s, (0, 0, 0), (1, 1, 1))
    keys, ss, _ = oracle.sort_by_key(keys, s)
    ss = np.ascontiguousarray(ss)
    dx = oracle.deltas_xor(keys)
    nodes, leaves, root, _ = oracle.albvh(ss, dx, 16)
    tree = gh.Tree(n, 16, device=cuda)
    gh.ALBVH_sph(_dev(ss, cuda), _dev(dx.view(np.int32), cuda), tree)
    assert np.array_equal(tree.leaves.cpu().numpy(), leaves)
    assert np.array_equal(tree.nodes.cpu().numpy(), nodes)
    assert int(tree.root_index.item()) == root


def test_albvh_rejects_small_input(gh, oracle, cuda):
    s = _spheres(oracle, 32)
    tree = gh.Tree(32, 32, device=cuda)
    with pytest.raises(ValueError):   # std::invalid_argument, albvh.cuh:795-799
        gh.build_tree(_dev(s, cuda), tree, (0, 0, 0), (1, 1, 1))


def _rays_from(origin, dirs, length):
    r = np.empty((len(dirs), 7), np.float32)
    r[:, 0:3] = dirs
    r[:, 3:6] = origin
    r[:, 6] = length
    return r


@pytest.fixture(params=[True, False], ids=["reorder", "caller-order"])
def ray_order(request, gh):
    gh.set_ray_reorder(request.param)
    yield request.param
    gh.set_ray_reorder(True)


@pytest.mark.parametrize("n,n_rays,mpl", [(20000, 1024, 32), (20000, 96, 1), (100000, 3200, 32)])
def test_trace_hitcounts_equal_brute_force(gh, oracle, cuda, ray_order, n, n_rays, mpl):
    """The reference's own criterion (tests/tree_traversal/tree_traversal.cu:65-100) on
    the hitcounts workload (tests/hitcounts/hitcounts.cu:47-58)."""
    s = _spheres(oracle, n)
    d, tree, ss, nodes, leaves, root = _build_both(gh, oracle, cuda, s, mpl)
    rays = gh.uniform_random_rays(n_rays, (0.5, 0.5, 0.5), 2.0, seed=1234, device=cuda)
    hc = torch.empty(n_rays, dtype=torch.int32, device=cuda)
    gh.trace_hitcounts_sph(rays, d, tree, hc)
    gh.trace_status()
    ref = oracle.brute_hitcounts(rays.cpu().numpy(), ss)
    assert np.array_equal(hc.cpu().numpy(), ref)
    assert ref.sum() > 0


def test_trace_tree_traversal_config(gh, oracle, cuda):
    """tests/tree_traversal geometry: spheres U([-1e4,1e4]^3), r in [80,400), rays from 0."""
    n, n_rays = 50000, 640
    s = _spheres(oracle, n, (-1e4, -1e4, -1e4, 80.0), (1e4, 1e4, 1e4, 400.0))
    d, tree, ss, *_ = _build_both(gh, oracle, cuda, s, 32, (-1e4,) * 3, (1e4,) * 3)
    rays = gh.uniform_random_rays(n_rays, (0.0, 0.0, 0.0), 2e4, seed=7, device=cuda)
    hc = torch.empty(n_rays, dtype=torch.int32, device=cuda)
    gh.trace_hitcounts_sph(rays, d, tree, hc)
    assert np.array_equal(hc.cpu().numpy(), oracle.brute_hitcounts(rays.cpu().numpy(), ss))


def test_trace_cumulative_bitexact_and_tolerance(gh, oracle, cuda, ray_order, integral_mode):
    n, n_side = 60000, 32
    s = _spheres(oracle, n, (0, 0, 0, 0.01), (1, 1, 1, 0.05))
    d, tree, ss, *_ = _build_both(gh, oracle, cuda, s, 32)
    rays, area = gh.orthogonal_rays_z(n_side, (0, 0, 0, 0), (1, 1, 1, 0), device=cuda)
    out = torch.empty(len(rays), dtype=torch.float32, device=cuda)
    gh.trace_cumulative_sph(rays, d, tree, out)
    ref32, ref64 = oracle.brute_cumulative(rays.cpu().numpy(), ss)
    check_column_densities(out.cpu().numpy(), ref32, ref64, integral_mode)


@pytest.mark.parametrize("axis,sense", [(0, 1), (0, -1), (1, 1), (1, -1), (2, 1), (2, -1)])
def test_axis_aligned_packets_partial_segments(gh, oracle, cuda, ray_order, integral_mode, axis, sense):
    """Axis-aligned packets take their own sweep (exact beam cull, FMA dot product, rounds
    that skip the [0, length) range tests when every candidate is provably inside).  Rays that
    start and end INSIDE the particle box, with per-ray origins and lengths, on sphere centres'
    own coordinates (dot == 0 and dot == length ties), in both senses of all three axes."""
    n, side = 40000, 48
    s = _spheres(oracle, n, (0, 0, 0, 0.004), (1, 1, 1, 0.03))
    d, tree, ss, *_ = _build_both(gh, oracle, cuda, s, 32)
    rng = np.random.default_rng(100 * axis + sense + 7)
    u, v = np.meshgrid((np.arange(side) + 0.5) / side, (np.arange(side) + 0.5) / side)
    rays = np.zeros((side * side, 7), np.float32)
    perp = [k for k in range(3) if k != axis]
    rays[:, axis] = sense
    rays[:, 3 + perp[0]] = u.ravel()
    rays[:, 3 + perp[1]] = v.ravel()
    start = rng.uniform(0.1, 0.6, len(rays)).astype(np.float32)
    length = rng.uniform(0.05, 0.5, len(rays)).astype(np.float32)
    # ties: a third of the rays start exactly at, or end exactly at, a sphere centre's coordinate
    pick = ss[rng.integers(0, n, len(rays)), axis]
    third = np.arange(len(rays)) % 3
    start = np.where(third == 0, pick, start).astype(np.float32)
    length = np.where(third == 1, np.abs(pick - start), length).astype(np.float32)
    rays[:, 3 + axis] = start if sense > 0 else (1.0 - start).astype(np.float32)
    rays[:, 6] = np.maximum(length, np.float32(1e-3))
    dr = _dev(rays, cuda)
    hc = torch.empty(len(rays), dtype=torch.int32, device=cuda)
    cu = torch.empty(len(rays), dtype=torch.float32, device=cuda)
    gh.trace_hitcounts_sph(dr, d, tree, hc)
    gh.trace_cumulative_sph(dr, d, tree, cu)
    gh.trace_status()
    assert np.array_equal(hc.cpu().numpy(), oracle.brute_hitcounts(rays, ss))
    ref32, ref64 = oracle.brute_cumulative(rays, ss)
    got = cu.cpu().numpy()
    nz = ref64 > 0
    assert np.array_equal(got == 0, ~nz)
    # short segments: some rays see one grazing hit only -> absolute floor from the largest term
    max_term = float(oracle.kernel_table()[0]) / 0.004 ** 2
    check_column_densities(got[nz], ref32[nz], ref64[nz], integral_mode, max_term=max_term)


def test_orthogonal_rays_match_oracle(gh, oracle, cuda):
    rays, area = gh.orthogonal_rays_z(64, (-1, 0.5, 2, 0.1), (3, 1.5, 4, 0.2), device=cuda)
    ref, ref_area = oracle.orthogonal_rays_z(64, (-1, 0.5, 2, 0.1), (3, 1.5, 4, 0.2))
    assert area == ref_area
    assert np.array_equal(rays.cpu().numpy().view(np.uint32).ravel(),
                          ref.view(np.uint32).ravel() if ref.dtype == np.uint32 else
                          np.frombuffer(ref.tobytes(), np.uint32))


def test_trace_rejects_bad_ray_count(gh, oracle, cuda):
    s = _spheres(oracle, 1000)
    d, tree, *_ = _build_both(gh, oracle, cuda, s, 8)
    rays = gh.uniform_random_rays(33, (0.5, 0.5, 0.5), 2.0, device=cuda)
    with pytest.raises(ValueError):   # bintree_trace.cuh:231-238
        gh.trace_hitcounts_sph(rays, d, tree, torch.empty(33, dtype=torch.int32, device=cuda))


def test_hit_integral_arithmetic_bitexact(gh, oracle, cuda):
    """sqrt, 1/h, the fp64 lerp and the final products against the oracle on 4M inputs,
    including the edges: b2 = 0, denormal and tiny b2, b2 just below h^2, table clamp."""
    rng = np.random.default_rng(5)
    n = 1 << 22
    h = np.exp(rng.uniform(np.log(1e-6), np.log(1e6), n)).astype(np.float32)
    frac = rng.uniform(0, 1, n).astype(np.float32)
    b2 = (frac * h) ** 2
    b2 = b2.astype(np.float32)
    b2[:1000] = 0.0
    b2[1000:2000] = np.float32(1e-40)            # denormal
    b2[2000:3000] = rng.uniform(1e-38, 1e-28, 1000).astype(np.float32)
    b2[3000:4000] = np.nextafter((h[3000:4000] * h[3000:4000]).astype(np.float32), np.float32(0))
    b2[4000:5000] = (h[4000:5000] * h[4000:5000]).astype(np.float32)   # table clamp branch
    got = gh.hit_integrals(_dev(b2, cuda), _dev(h, cuda)).cpu().numpy()
    L = oracle.lib()
    import ctypes as C
    L.go_hit_integral_array.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    ref = np.empty(n, np.float32)
    L.go_hit_integral_array(b2.ctypes.data, h.ctypes.data, n, ref.ctypes.data)
    bad = np.nonzero(got.view(np.uint32) != ref.view(np.uint32))[0]
    assert len(bad) == 0, (bad[:5], b2[bad[:5]], h[bad[:5]], got[bad[:5]], ref[bad[:5]])


@pytest.mark.parametrize("n,n_rays", [(60000, 1024), (300000, 4096)])
def test_packet_split_does_not_change_results(gh, oracle, cuda, n, n_rays, integral_mode):
    """A packet walked by 1, 2, 4 or 8 waves (each owning 8/K of the summation classes)
    gives bit-identical column densities and hit counts, equal to the oracle's class-ordered
    sum / brute-force count."""
    s = _spheres(oracle, n, (0, 0, 0, 0.005), (1, 1, 1, 0.04))
    d, tree, ss, *_ = _build_both(gh, oracle, cuda, s, 32)
    rays = gh.uniform_random_rays(n_rays, (0.5, 0.5, 0.5), 2.0, seed=5, device=cuda)
    sub = np.linspace(0, n_rays - 1, 256).astype(np.int64)
    ref32, ref64 = oracle.brute_cumulative(rays.cpu().numpy()[sub], ss)
    refc = oracle.brute_hitcounts(rays.cpu().numpy()[sub], ss)
    base_sum = base_cnt = None
    try:
        for k in (1, 2, 4, 8, -1):
            gh.set_packet_split(k)
            out = torch.empty(n_rays, dtype=torch.float32, device=cuda)
            cnt = torch.empty(n_rays, dtype=torch.int32, device=cuda)
            gh.trace_cumulative_sph(rays, d, tree, out)
            gh.trace_hitcounts_sph(rays, d, tree, cnt)
            gh.trace_status()
            if base_sum is None:
                base_sum, base_cnt = out.clone(), cnt.clone()
                check_column_densities(out.cpu().numpy()[sub], ref32, ref64, integral_mode)
                assert np.array_equal(cnt.cpu().numpy()[sub], refc)
            assert torch.equal(out.view(torch.int32), base_sum.view(torch.int32)), k
            assert torch.equal(cnt, base_cnt), k
    finally:
        gh.set_packet_split(-1)
