"""GPU parity, part 2: per-hit trace, scans, ray generators, the committed fixtures, the C++
header mirror, and size-independent properties at BASELINE.json's full sizes."""
import json
import os
import subprocess

import numpy as np
import pytest
import torch
from conftest import check_column_densities

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def _build(gh, cuda, s, mpl, low=(0, 0, 0), high=(1, 1, 1)):
    d = _dev(s, cuda)
    tree = gh.Tree(len(s), mpl, device=cuda)
    gh.build_tree(d, tree, low, high)
    return d, tree


# ---- committed fixtures -------------------------------------------------------------------
def test_golden_pipeline_on_gpu(gh, cuda, integral_mode):
    g = np.load(os.path.join(GOLD, "pipeline_n4096.npz"))
    s = g["spheres"]
    d = _dev(s, cuda)
    k30 = torch.empty(len(s), dtype=torch.int32, device=cuda)
    k63 = torch.empty(len(s), dtype=torch.int64, device=cuda)
    gh.morton_keys_sph(d, k30, (0, 0, 0), (1, 1, 1))
    gh.morton_keys_sph(d, k63, (0, 0, 0), (1, 1, 1))
    assert np.array_equal(k30.cpu().numpy().view(np.uint32), g["keys30"])
    assert np.array_equal(k63.cpu().numpy().view(np.uint64), g["keys63"])
    perm = gh.sort_by_key(k30, d, 0, 30, want_perm=True)
    assert np.array_equal(perm.cpu().numpy().view(np.uint32), g["order"])
    dl = torch.empty(len(s) + 1, dtype=torch.float32, device=cuda)
    gh.euclidean_deltas_sph(d, dl)
    assert np.array_equal(dl.cpu().numpy().view(np.uint32), g["deltas"].view(np.uint32))
    for mpl in (1, 8, 32):
        tree = gh.Tree(len(s), mpl, device=cuda)
        gh.ALBVH_sph(d, dl, tree)
        assert np.array_equal(tree.nodes.cpu().numpy(), g["nodes_%d" % mpl])
        assert np.array_equal(tree.leaves.cpu().numpy(), g["leaves_%d" % mpl])
        assert int(tree.root_index.item()) == int(g["root_%d" % mpl])
        rays = _dev(g["rays"], cuda)
        hc = torch.empty(len(rays), dtype=torch.int32, device=cuda)
        gh.trace_hitcounts_sph(rays, d, tree, hc)
        assert np.array_equal(hc.cpu().numpy(), g["hit_counts"])
        cu = torch.empty(len(rays), dtype=torch.float32, device=cuda)
        gh.trace_cumulative_sph(rays, d, tree, cu)
        check_column_densities(cu.cpu().numpy(), g["cumulative32"], g["cumulative64"], integral_mode)
    so = _dev(g["seg_offsets"], cuda); sd = _dev(g["seg_data"], cuda)
    out = torch.empty_like(sd)
    gh.exclusive_segmented_scan(so, sd, out)
    assert np.array_equal(out.cpu().numpy(), g["seg_result"])


def test_healpix_rays_against_reference_run(gh, cuda):
    ref = np.load(os.path.join(GOLD, "healpix_nside4_ref.npy"))     # reference chealpix output
    rays = gh.healpix_rays(4, (0.5, 0.25, 0.125), 3.0, device=cuda).cpu().numpy()
    assert rays.shape == (192, 7)
    # device cos/sin/sqrt in fp64, then rounded to fp32: at most 1 ulp from the reference
    assert np.allclose(rays[:, :3], ref.astype(np.float32), rtol=0, atol=1.2e-7)
    assert np.all(rays[:, 3:6] == np.array([0.5, 0.25, 0.125], np.float32)) and np.all(rays[:, 6] == 3.0)


def test_isotropic_rays_are_unit_sorted_and_reproducible(gh, oracle, cuda):
    a = gh.uniform_random_rays(4096, (0, 0, 0), 1.0, seed=11, device=cuda).cpu().numpy()
    b = gh.uniform_random_rays(4096, (0, 0, 0), 1.0, seed=11, device=cuda).cpu().numpy()
    c = gh.uniform_random_rays(4096, (0, 0, 0), 1.0, seed=12, device=cuda).cpu().numpy()
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert np.allclose(np.linalg.norm(a[:, :3], axis=1), 1.0, atol=3e-7)
    keys = oracle.ray_dir_keys(a)                     # gen_rays.cuh:38-43
    assert np.all(np.diff(keys.astype(np.int64)) >= 0)
    assert np.abs(a[:, :3].mean(axis=0)).max() < 0.05   # isotropy, first moment


@pytest.mark.parametrize("octant", [7, 5, 2, 0])
def test_single_octant_rays(gh, oracle, cuda, octant):
    """uniform_random_rays_single_octant (gen_rays.cuh:62-97): component signs per enum Octants
    (bit 2 = x, 1 = y, 0 = z; set = positive), unit length, direction-key order, and the
    octant-restricted first moment of an isotropic distribution (1/2 per component)."""
    r = gh.uniform_random_rays_single_octant(8192, (1, 2, 3), 5.0, octant, seed=9, device=cuda).cpu().numpy()
    sign = np.array([1 if octant & 4 else -1, 1 if octant & 2 else -1, 1 if octant & 1 else -1])
    assert np.all(r[:, :3] * sign >= 0)
    assert np.allclose(np.linalg.norm(r[:, :3], axis=1), 1.0, atol=3e-7)
    assert np.all(r[:, 3:6] == np.array([1, 2, 3], np.float32)) and np.all(r[:, 6] == 5.0)
    assert np.all(np.diff(oracle.ray_dir_keys(r).astype(np.int64)) >= 0)
    assert np.abs(r[:, :3].mean(axis=0) * sign - 0.5).max() < 0.02
    same = gh.uniform_random_rays_single_octant(8192, (1, 2, 3), 5.0, octant, seed=9, device=cuda).cpu().numpy()
    assert np.array_equal(r, same)
    with pytest.raises(Exception):
        gh.uniform_random_rays_single_octant(64, (0, 0, 0), 1.0, 8, device=cuda)


@pytest.mark.parametrize("dtype,cols", [(np.float32, 3), (np.float32, 4), (np.float64, 4)])
def test_one_to_many_rays(gh, oracle, cuda, dtype, cols):
    """one_to_many_rays (gen_rays.cuh:99-208): unsorted == the oracle's restatement bit for bit;
    DirectionSort / EndPointSort == a stable sort of the unsorted rays by ray_dir_morton_key /
    by the end points' 30-bit Morton keys; an unknown sort type is an invalid argument."""
    rng = np.random.default_rng(5)
    pts = rng.uniform(-2, 3, (5000, cols)).astype(dtype)
    origin = (0.25, -0.5, 1.0)
    d = _dev(pts, cuda)
    ref = oracle.one_to_many_rays(origin, pts)
    plain = gh.one_to_many_rays(origin, d, gh.NoSort).cpu().numpy()
    refa = ref.view(np.float32).reshape(-1, 7)
    assert np.array_equal(plain.view(np.uint32), refa.view(np.uint32))
    assert np.allclose(plain[:, 3:6] + plain[:, :3] * plain[:, 6:7], pts[:, :3], atol=2e-6)
    by_dir = gh.one_to_many_rays(origin, d, gh.DirectionSort).cpu().numpy()
    order = np.argsort(oracle.ray_dir_keys(ref), kind="stable")
    assert np.array_equal(by_dir, plain[order])
    bot, top = pts[:, :3].min(0).astype(np.float32), pts[:, :3].max(0).astype(np.float32)
    by_end = gh.one_to_many_rays(origin, d, gh.EndPointSort, bot, top).cpu().numpy()
    p4 = np.zeros((len(pts), 4), np.float32); p4[:, :3] = pts[:, :3].astype(np.float32)
    order = np.argsort(oracle.morton_keys30(p4, bot, top), kind="stable")
    assert np.array_equal(by_end, plain[order])
    with pytest.raises(Exception):
        gh.one_to_many_rays(origin, d, 3)


def test_plane_parallel_random_rays(gh, cuda):
    """plane_parallel_random_rays (gen_rays.cuh:210-262): one ray per cell, origin inside its
    cell of the (w, h) grid on the base plane, direction normalize(cross(w, h)); the comment's
    own example (base (5,0,10), w (-5,0,0), h (0,6,0) => direction (0,0,-1))."""
    W, H = 40, 30
    base, w, h = (5.0, 0.0, 10.0), (-5.0, 0.0, 0.0), (0.0, 6.0, 0.0)
    r = gh.plane_parallel_random_rays(W, H, base, w, h, 7.5, seed=3, device=cuda).cpu().numpy()
    assert r.shape == (W * H, 7)
    assert np.all(r[:, :3] == np.array([0, 0, -1], np.float32)) and np.all(r[:, 6] == 7.5)
    i, j = np.arange(W * H) % W, np.arange(W * H) // W
    fx = (5.0 - r[:, 3]) / (5.0 / W)            # cells along w run from x = 5 downwards
    fy = r[:, 4] / (6.0 / H)
    assert np.all((fx >= i - 1e-4) & (fx <= i + 1 + 1e-4))
    assert np.all((fy >= j - 1e-4) & (fy <= j + 1 + 1e-4)) and np.all(r[:, 5] == 10.0)
    frac = np.concatenate([fx - i, fy - j])
    assert abs(frac.mean() - 0.5) < 0.02 and frac.std() > 0.25    # uniform within the cell
    again = gh.plane_parallel_random_rays(W, H, base, w, h, 7.5, seed=3, device=cuda).cpu().numpy()
    other = gh.plane_parallel_random_rays(W, H, base, w, h, 7.5, seed=4, device=cuda).cpu().numpy()
    assert np.array_equal(r, again) and not np.array_equal(r, other)
    # an oblique plane: every origin satisfies dot(o - base, n) = 0
    r2 = gh.plane_parallel_random_rays(16, 16, (1, 1, 1), (1, 2, 0), (0, 1, 3), 1.0, device=cuda).cpu().numpy()
    n = np.cross((1, 2, 0), (0, 1, 3)); n = n / np.linalg.norm(n)
    assert np.allclose(r2[:, :3], n, atol=1e-6)
    assert np.abs((r2[:, 3:6] - 1.0) @ n).max() < 1e-5


def test_orthographic_projection_rays(gh, oracle, cuda):
    """orthographic_projection_rays (gen_rays.cuh:264-329) against the oracle's restatement
    (bit for bit), and against orthogonal_rays_z for the -z view of a unit box."""
    args = (96, 64, (0.5, -2.0, 0.25), (0.1, 0.4, 0.3), (0.0, 0.2, 1.0), 1.5, 9.0)
    got = gh.orthographic_projection_rays(*args, device=cuda).cpu().numpy()
    ref = oracle.orthographic_projection_rays(*args).view(np.float32).reshape(-1, 7)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    d = got[0, :3]
    assert np.allclose(np.linalg.norm(d), 1.0, atol=1e-6) and np.all(got[:, :3] == d)
    assert np.abs((got[:, 3:6] - np.array(args[2], np.float32)) @ d).max() < 1e-5   # in the image plane
    # tests/helper/rays.cuh:55-79 is this generator with a -z camera above the box
    lo, hi = (0, 0, 0, 0), (1, 1, 1, 0)
    zr, _ = gh.orthogonal_rays_z(32, lo, hi, device=cuda)
    pr = gh.orthographic_projection_rays(32, 32, (0.5, 0.5, 1.0), (0.5, 0.5, 0.0), (0, 1, 0), 1.0,
                                         2.0, device=cuda)
    assert np.array_equal(zr.cpu().numpy(), pr.cpu().numpy())


@pytest.mark.parametrize("dtype,cols,bits", [(np.float64, 4, 63), (np.float64, 4, 30),
                                             (np.float32, 3, 63), (np.float64, 3, 30)])
def test_morton_keys_of_double4_and_float3_points(gh, oracle, cuda, dtype, cols, bits):
    """tests/morton_key_kernel/63bit_keys.cu: keys of double4 points with float3 bounds
    (-1, 1).  The device narrows each co-ordinate to float first (CentroidSphere returns
    float3) and then applies scale * (c - min), scale = span / (top - bot) in float
    (kernels/morton.cuh:38-47,104-113) -- restated by the oracle on the narrowed points.
    (That test's own host loop multiplies by MAX_KEY without dividing by the box width, so it
    cannot serve as the oracle.)"""
    n = 10000
    rng = np.random.default_rng(63)
    pts = rng.uniform(-1, 1, (n, cols)).astype(dtype)
    bot, top = np.float32([-1, -1, -1]), np.float32([1, 1, 1])
    keys = torch.empty(n, dtype=torch.int64 if bits == 63 else torch.int32, device=cuda)
    gh.morton_keys_points(_dev(pts, cuda), keys, bot, top)
    p4 = np.zeros((n, 4), np.float32); p4[:, :3] = pts[:, :3].astype(np.float32)
    if bits == 63:
        assert np.array_equal(keys.cpu().numpy().view(np.uint64), oracle.morton_keys63(p4, bot, top))
    else:
        assert np.array_equal(keys.cpu().numpy().view(np.uint32), oracle.morton_keys30(p4, bot, top))
    if dtype == np.float64 and cols == 4 and bits == 63:
        # morton_keys63_sort_sph<double4> (build_sph.cuh:65-82): sort the 32-byte records by key
        d = _dev(pts, cuda)
        gh.sort_by_key(keys, d, 0, 63)
        order = np.argsort(oracle.morton_keys63(p4, bot, top), kind="stable")
        assert np.array_equal(d.cpu().numpy(), pts[order])


def test_isotropic_ray_statistics(gh, cuda):
    """tests/isotropic_ray_stats/uniformity_stats.cu: Rayleigh's z = 3 R^2 / n <= 7.815 and Gine's
    F_n = A_n + G_n <= 1.9478, A_n = n - 4/(n pi) SUM psi_ij (Beran), G_n = n/2 - 4/(n pi) SUM
    sin psi_ij over all pairs, for the default-seed isotropic generator (4096 rays)."""
    n = 4096
    r = gh.uniform_random_rays(n, (0.0, 0.0, 0.0), 1.0, seed=1234, device=cuda)[:, :3].double()
    R2 = float((r.sum(dim=0) ** 2).sum())
    z = 3.0 * R2 / n
    cosm = (r @ r.T).clamp(-1.0, 1.0)
    iu = torch.triu_indices(n, n, offset=1, device=cuda)
    c = cosm[iu[0], iu[1]]
    # angular separation with the numerically safe form 2 asin(|a - b| / 2)
    diff = (r[iu[0]] - r[iu[1]]).norm(dim=1)
    psi = 2.0 * torch.asin((diff / 2.0).clamp(max=1.0))
    coeff = 4.0 / (n * np.pi)
    An = n - coeff * float(psi.sum())
    Gn = n / 2.0 - coeff * float(torch.sin(psi).sum())
    assert c.numel() == n * (n - 1) // 2
    assert z <= 7.815, z
    assert An + Gn <= 1.9478, (An, Gn)


# ---- trace_sph (two-pass per-hit output) ------------------------------------------------------
@pytest.mark.parametrize("n,n_rays,mpl", [(30000, 512, 32), (5000, 64, 1)])
def test_trace_sph_hits(gh, oracle, cuda, n, n_rays, mpl):
    s = oracle.random_real4(n, (0, 0, 0, 0), (1, 1, 1, 0.08))
    d, tree = _build(gh, cuda, s, mpl)
    rays = gh.uniform_random_rays(n_rays, (0.5, 0.5, 0.5), 2.0, seed=3, device=cuda)
    offs, idx, integ, dist = gh.trace_sph(rays, d, tree)
    gh.trace_status()
    ro, ri, rw, rd = oracle.brute_hits(rays.cpu().numpy(), d.cpu().numpy())
    assert np.array_equal(offs.cpu().numpy(), ro)
    assert np.array_equal(idx.cpu().numpy(), ri)           # ascending primitive index per ray
    assert np.array_equal(integ.cpu().numpy().view(np.uint32), rw.view(np.uint32))
    assert np.array_equal(dist.cpu().numpy().view(np.uint32), rd.view(np.uint32))
    assert np.all(dist.cpu().numpy() >= 0)                  # tests/distance_sort: none < 0
    # per-ray exclusive scan of the integrals == optical depth in front of each hit
    out = torch.empty_like(integ)
    gh.exclusive_segmented_scan(offs, integ, out)
    assert np.allclose(out.cpu().numpy(), oracle.segscan(ro, rw), rtol=2e-6, atol=1e-30)


def test_trace_with_sentinels(gh, oracle, cuda):
    s = oracle.random_real4(20000, (0, 0, 0, 0), (1, 1, 1, 0.08))
    d, tree = _build(gh, cuda, s, 16)
    rays = gh.uniform_random_rays(256, (0.5, 0.5, 0.5), 2.0, seed=9, device=cuda)
    offs, idx, integ, dist = gh.trace_with_sentinels_sph(rays, d, tree, -7, -1.5, 1e30)
    ro, ri, rw, rd = oracle.brute_hits(rays.cpu().numpy(), d.cpu().numpy())
    counts = np.diff(np.concatenate([ro, [len(ri)]]))
    so = ro + np.arange(len(ro), dtype=np.int32)            # trace_sph.cuh:205-208
    assert np.array_equal(offs.cpu().numpy(), so)
    gi = idx.cpu().numpy(); gw = integ.cpu().numpy(); gd = dist.cpu().numpy()
    assert len(gi) == len(ri) + len(ro)
    for r in (0, 1, 17, 255):
        a, c = so[r], counts[r]
        assert np.array_equal(gi[a:a + c], ri[ro[r]:ro[r] + c])
        assert np.array_equal(gw[a:a + c].view(np.uint32), rw[ro[r]:ro[r] + c].view(np.uint32))
        assert gi[a + c] == -7 and gw[a + c] == np.float32(-1.5) and gd[a + c] == np.float32(1e30)
    sent = so + counts
    assert np.all(gi[sent] == -7)


def test_sort_by_distance(gh, oracle, cuda):
    """tests/distance_sort/distance_sort.cu:22-79,125-130: after trace_sph + sort_by_distance
    distances are non-decreasing within each ray and none is negative; here also equal to a
    stable per-segment host sort, with indices and integrals following."""
    s = oracle.random_real4(30000, (0, 0, 0, 0), (1, 1, 1, 0.08))
    d, tree = _build(gh, cuda, s, 32)
    rays = gh.uniform_random_rays(640, (0.5, 0.5, 0.5), 2.0, seed=21, device=cuda)
    offs, idx, integ, dist = gh.trace_sph(rays, d, tree)
    o = offs.cpu().numpy(); i0 = idx.cpu().numpy(); w0 = integ.cpu().numpy(); d0 = dist.cpu().numpy()
    gh.sort_by_distance(dist, offs, idx, integ)
    i1 = idx.cpu().numpy(); w1 = integ.cpu().numpy(); d1 = dist.cpu().numpy()
    assert np.all(d1 >= 0)
    ends = np.concatenate([o[1:], [len(d0)]])
    for r in range(len(o)):
        a, b = o[r], ends[r]
        order = np.argsort(d0[a:b], kind="stable")
        assert np.array_equal(d1[a:b].view(np.uint32), d0[a:b][order].view(np.uint32))
        assert np.array_equal(i1[a:b], i0[a:b][order])
        assert np.array_equal(w1[a:b].view(np.uint32), w0[a:b][order].view(np.uint32))
        assert np.all(np.diff(d1[a:b]) >= 0)


@pytest.mark.parametrize("case", ["many-short", "few-long", "huge"])
def test_sort_by_distance_random_segments(gh, cuda, case):
    """Both implementations behind sort_by_distance: one wavefront per segment (mean segment
    length <= 32768, here with segments from 0 to 40 000 elements, negative zeros and negative
    distances) and the composite-key global sort ("huge": mean length above that)."""
    rng = np.random.default_rng(4)
    if case == "many-short":
        sizes = rng.integers(0, 300, 5000); sizes[::9] = 0
    elif case == "few-long":
        sizes = np.array([40000, 0, 1, 2, 63, 64, 65, 12345, 0, 30000, 7])
    else:
        sizes = np.array([70000, 0, 90000])
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
    n = int(sizes.sum())
    dist = rng.integers(0, 50, n).astype(np.float32) * 0.25     # many ties
    if case != "many-short":
        dist = (rng.integers(-2000, 2000, n).astype(np.float32) * 0.125)
        dist[rng.integers(0, n, 50)] = -0.0
    idx = np.arange(n, dtype=np.int32); data = rng.standard_normal(n).astype(np.float32)
    dd = _dev(dist, cuda); di = _dev(idx, cuda); dw = _dev(data, cuda)
    gh.sort_by_distance(dd, _dev(offs, cuda), di, dw)
    seg = np.repeat(np.arange(len(sizes)), sizes)
    order = np.lexsort((np.arange(n), dist, seg))                 # stable within segments
    assert np.array_equal(dd.cpu().numpy().view(np.uint32), dist[order].view(np.uint32))
    assert np.array_equal(di.cpu().numpy(), idx[order])
    assert np.array_equal(dw.cpu().numpy(), data[order])


# ---- scans ---------------------------------------------------------------------------------
@pytest.mark.parametrize("count,max_seg", [(1, 1), (1000, 5), (10000, 3), (100000, 64),
                                           (1 << 20, 2000), (3000000, 100000)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_segmented_scan_matches_host(gh, oracle, cuda, count, max_seg, dtype):
    """tests/segmented_scan/segmented_scan.cu:65-162: random segment sizes incl. empties,
    integer-valued data 1..9 (exact in fp), GPU == sequential host scan."""
    rng = np.random.default_rng(count + max_seg)
    sizes = []
    total = 0
    while total < count:
        sz = int(rng.integers(0, min(max_seg, count - total) + 1))
        sizes.append(sz); total += sz
    sizes = np.array(sizes, np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
    data = rng.integers(1, 10, count).astype(dtype)
    so = _dev(offs, cuda); sd = _dev(data, cuda)
    out = torch.empty_like(sd)
    gh.exclusive_segmented_scan(so, sd, out)
    assert np.array_equal(out.cpu().numpy(), oracle.segscan(offs, data))
    gh.exclusive_segmented_scan(so, sd, sd)          # in place, as scan.cuh:13 allows
    assert np.array_equal(sd.cpu().numpy(), oracle.segscan(offs, data))


@pytest.mark.parametrize("n", [1, 63, 8192, 8193, 1 << 20, 5000001])
def test_exclusive_scan(gh, oracle, cuda, n):
    a = np.random.default_rng(n).integers(0, 400, n).astype(np.int32)   # total < 2^31
    d = _dev(a, cuda)
    total = gh.exclusive_scan(d, d)
    ref, rt = oracle.exclusive_scan_i32(a)
    assert total == rt and np.array_equal(d.cpu().numpy(), ref)


def test_weighted_segmented_scan(gh, oracle, cuda):
    rng = np.random.default_rng(9)
    n = 50000
    x = rng.integers(1, 5, n).astype(np.float32); w = rng.integers(1, 4, 100).astype(np.float32)
    m = rng.integers(0, 100, n).astype(np.int32)
    offs = np.arange(0, n, 500, dtype=np.int32)
    out = torch.empty(n, dtype=torch.float32, device=cuda)
    gh.weighted_exclusive_segmented_scan(_dev(x, cuda), _dev(w, cuda), _dev(m, cuda), _dev(offs, cuda), out)
    assert np.array_equal(out.cpu().numpy(), oracle.segscan(offs, w[m] * x))


def test_trace_sph_staged_outputs_large_batch(gh, oracle, cuda):
    """With >= 4096 packets the per-hit trace stages its outputs in LDS and drains them eight
    entries at a time: same offsets / indices / integrals / distances as the oracle (checked on
    400 rays of 262 176, a ray count that leaves a half-empty last packet)."""
    n, n_rays = 30000, 4096 * 64 + 32
    s = oracle.random_real4(n, (0, 0, 0, 0.002), (1, 1, 1, 0.03))
    d, tree = _build(gh, cuda, s, 32)
    rays = gh.uniform_random_rays(n_rays, (0.5, 0.5, 0.5), 2.0, seed=21, device=cuda)
    offs, idx, integ, dist = gh.trace_sph(rays, d, tree)
    gh.trace_status()
    offs = offs.cpu().numpy(); idx = idx.cpu().numpy(); integ = integ.cpu().numpy(); dist = dist.cpu().numpy()
    hc = torch.empty(n_rays, dtype=torch.int32, device=cuda)
    gh.trace_hitcounts_sph(rays, d, tree, hc)
    hc = hc.cpu().numpy()
    assert np.array_equal(offs, np.concatenate([[0], np.cumsum(hc)[:-1]])) and len(idx) == hc.sum()
    sub = np.unique(np.concatenate([np.linspace(0, n_rays - 1, 390).astype(np.int64),
                                    np.arange(n_rays - 10, n_rays)]))
    ro, ri, rw, rd = oracle.brute_hits(rays.cpu().numpy()[sub], d.cpu().numpy())
    for k, r in enumerate(sub):
        a, b = offs[r], offs[r] + hc[r]
        ra, rb = ro[k], (ro[k + 1] if k + 1 < len(sub) else len(ri))
        assert b - a == rb - ra
        assert np.array_equal(idx[a:b], ri[ra:rb])
        assert np.array_equal(integ[a:b].view(np.uint32), rw[ra:rb].view(np.uint32))
        assert np.array_equal(dist[a:b].view(np.uint32), rd[ra:rb].view(np.uint32))


def test_sphere_intersection_against_exact_arithmetic(gh, oracle, cuda):
    """tests/sphere_intersection/sphere_intersection.cu: every (ray, sphere) hit decision against
    an exact reference -- does a*t^2 + b*t + c = 0 have a root in [0, length], in rational
    arithmetic on the inputs as given (the direction is NOT assumed to be of unit length)?
    Disagreement is allowed only where |1 - b_ref^2 / R^2| <= 1e-8 (the reference's
    tolerance).  Same geometry: centres U[-1e4, 1e4)^3, radii U[80, 400), pushed 400 away from
    the rays' common origin so that no sphere contains it.  float64 screens the 5e7 pairs;
    every mismatch is then decided with fractions.Fraction."""
    from fractions import Fraction
    n, n_rays = 50000, 1024
    s = oracle.random_real4(n, (-1e4, -1e4, -1e4, 80.0), (1e4, 1e4, 1e4, 400.0))
    s[:, :3] += np.float32(400.0) * np.sign(s[:, :3])            # expand_functor, :22-36
    d, tree = _build(gh, cuda, s, 32, (-1.1e4,) * 3, (1.1e4,) * 3)
    rays = gh.uniform_random_rays(n_rays, (0.0, 0.0, 0.0), 2e4, seed=1234, device=cuda)
    offs, idx, _, _ = gh.trace_sph(rays, d, tree)
    gh.trace_status()
    offs = offs.cpu().numpy(); idx = idx.cpu().numpy()
    ss = d.cpu().numpy().astype(np.float64); rr = rays.cpu().numpy().astype(np.float64)
    hit_gpu = np.zeros((n_rays, n), bool)
    ends = np.concatenate([offs[1:], [len(idx)]])
    for r in range(n_rays):
        hit_gpu[r, idx[offs[r]:ends[r]]] = True
    R2 = ss[:, 3] ** 2
    n_checked = 0
    for r0 in range(0, n_rays, 64):                               # float64 screen, 64 rays at a time
        dd = rr[r0:r0 + 64, None, :3]; oo = rr[r0:r0 + 64, None, 3:6]; L = rr[r0:r0 + 64, None, 6]
        p = oo - ss[None, :, :3]
        a = (dd * dd).sum(-1); b = 2 * (dd * p).sum(-1); c = (p * p).sum(-1) - R2[None, :]
        disc = b * b - 4 * a * c
        tv = -b / (2 * a)
        fL = a * L * L + b * L + c
        ref = (disc >= 0) & ((c * fL <= 0) | ((tv >= 0) & (tv <= L) & (c >= 0) & (fL >= 0)))
        for rl, si in zip(*np.nonzero(ref != hit_gpu[r0:r0 + 64])):
            ray = [Fraction(float(x)) for x in rays.cpu().numpy()[r0 + rl]]
            sph = [Fraction(float(x)) for x in d.cpu().numpy()[si]]
            dx, dy, dz, ox, oy, oz, Lx = ray
            px, py, pz = ox - sph[0], oy - sph[1], oz - sph[2]
            A = dx * dx + dy * dy + dz * dz
            B = 2 * (dx * px + dy * py + dz * pz)
            Cc = px * px + py * py + pz * pz - sph[3] * sph[3]
            D = B * B - 4 * A * Cc
            f0, fl, tvx = Cc, A * Lx * Lx + B * Lx + Cc, -B / (2 * A)
            exact = D >= 0 and (f0 * fl <= 0 or (0 <= tvx <= Lx and f0 >= 0 and fl >= 0))
            b2 = (px * px + py * py + pz * pz) - (B / 2) ** 2 / A      # exact squared impact parameter
            n_checked += 1
            assert exact == bool(hit_gpu[r0 + rl, si]) or abs(1 - float(b2 / (sph[3] * sph[3]))) <= 1e-8, \
                (r0 + rl, si, exact, float(b2), float(sph[3] ** 2))
    assert hit_gpu.sum() > 10000 and n_checked < 200        # only boundary cases may need the exact path


# ---- KATs on the GPU -------------------------------------------------------------------------
def test_volume_integral_kat_on_gpu(gh, cuda):
    """tests/integrate/integrate.cu: two spheres (max_per_leaf 1), normalised volume
    integral == 1 +- 5e-4."""
    tol = json.load(open(os.path.join(GOLD, "kat.json")))["integrate_tolerance"]
    s = torch.tensor([[-0.5, -0.5, -0.5, 0.2], [0.5, 0.5, 0.5, 0.2]], dtype=torch.float32, device=cuda)
    tree = gh.Tree(2, 1, device=cuda)
    gh.build_tree(s, tree, (-1, -1, -1), (1, 1, 1))
    rays, area = gh.orthogonal_rays_z(512, (-1, -1, -1, 0.2), (1, 1, 1, 0.2), device=cuda)
    out = torch.empty(len(rays), dtype=torch.float32, device=cuda)
    gh.trace_cumulative_sph(rays, s, tree, out)
    integral = float(out.sum()) * area / 2
    assert abs(1.0 - integral) < tol


def test_integrate_gadget_like_kat(gh, cuda):
    """tests/integrate_gadget: every particle's kernel integrates to one, so the normalised
    volume integral over a snapshot is 1 (synthetic 32^3 jittered lattice)."""
    g = torch.Generator(device=cuda); g.manual_seed(42)
    n_side = 32
    n = n_side ** 3
    grid = torch.stack(torch.meshgrid(*[torch.arange(n_side, device=cuda)] * 3, indexing="ij"), -1)
    pos = (grid.reshape(-1, 3).float() + torch.rand((n, 3), generator=g, device=cuda)) / n_side
    h = (3 * 48 / (4 * np.pi * n)) ** (1 / 3)
    s = torch.cat([pos, torch.full((n, 1), h, device=cuda)], 1).contiguous()
    lo, hi = gh.min_max_vec4(s)
    tree = gh.Tree(n, 32, device=cuda)
    gh.build_tree(s, tree, lo[:3], hi[:3])
    rays, area = gh.orthogonal_rays_z(512, lo, hi, device=cuda)   # w = max h pads the grid
    out = torch.empty(len(rays), dtype=torch.float32, device=cuda)
    gh.trace_cumulative_sph(rays, s, tree, out)
    assert abs(1.0 - float(out.double().sum()) * area / n) < 5e-4


# ---- the C++ header mirror, run on the GPU -------------------------------------------------------
def test_cpp_tree_traversal_program(tmp_path):
    lib = os.path.join(ROOT, "grace-devel_amd", "lib")
    exe = tmp_path / "tree_traversal"
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "tree_traversal.cpp"), "-o", str(exe),
                           "-L" + lib, "-lgrace_hip", "-L" + os.path.join(ROOT, "oracle"),
                           "-lgrace_oracle", "-Wl,-rpath," + lib,
                           "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    r = subprocess.run([str(exe), "100000", "50", "32"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PASSED" in r.stdout, r.stdout + r.stderr


# ---- full BASELINE sizes: properties that need no brute force ------------------------------------
def test_config2_full_size_properties(gh, oracle, cuda, integral_mode):
    """LBVH build + hitcounts at 10^6 spheres / 10^5 rays (tests/hitcounts/hitcounts.cu):
    sortedness and stability of the build sort, leaf partition, packet-order independence
    (bitwise), per-hit sums vs cumulative, and exactness against the brute-force oracle on
    an evenly spaced 512-ray subset."""
    n, n_rays = 1000000, 100000 // 64 * 64 + 32     # = 100000 = 3125 * 32, not a multiple of 64
    assert n_rays == 100000
    s = oracle.random_real4(n, (0, 0, 0, 0), (1, 1, 1, 0.1))
    d = _dev(s, cuda)
    keys = torch.empty(n, dtype=torch.int32, device=cuda)
    gh.morton_keys_sph(d, keys, (0, 0, 0), (1, 1, 1))
    k0 = keys.clone()
    perm = gh.sort_by_key(keys, d, 0, 30, want_perm=True)
    kk = keys.cpu().numpy().astype(np.int64); pp = perm.cpu().numpy().astype(np.int64)
    assert np.all(np.diff(kk) >= 0)                                   # sorted
    assert np.all((np.diff(kk) > 0) | (np.diff(pp) > 0))              # stable
    assert np.array_equal(np.sort(pp), np.arange(n))                  # a permutation
    assert np.array_equal(k0.cpu().numpy()[pp], keys.cpu().numpy())
    keys2 = keys.clone(); gh.sort_by_key(keys2, None, 0, 30)          # idempotent
    assert torch.equal(keys2, keys)
    dl = torch.empty(n + 1, dtype=torch.float32, device=cuda)
    gh.euclidean_deltas_sph(d, dl)
    tree = gh.Tree(n, 32, device=cuda)
    gh.ALBVH_sph(d, dl, tree)
    lv = tree.leaves.cpu().numpy()
    assert lv[0, 0] == 0 and lv[:, 1].sum() == n and lv[:, 1].max() <= 32
    assert np.all(lv[1:, 0] == lv[:-1, 0] + lv[:-1, 1])
    nd = tree.nodes.cpu().numpy(); root = int(tree.root_index.item())
    assert nd[root, 2] == 0 and nd[root, 3] == len(lv) - 1

    rays = gh.uniform_random_rays(n_rays, (0.5, 0.5, 0.5), 2.0, seed=1234, device=cuda)
    hc = torch.empty(n_rays, dtype=torch.int32, device=cuda)
    cu = torch.empty(n_rays, dtype=torch.float32, device=cuda)
    gh.trace_hitcounts_sph(rays, d, tree, hc); gh.trace_cumulative_sph(rays, d, tree, cu)
    gh.set_ray_reorder(False)
    hc2 = torch.empty_like(hc); cu2 = torch.empty_like(cu)
    gh.trace_hitcounts_sph(rays, d, tree, hc2); gh.trace_cumulative_sph(rays, d, tree, cu2)
    gh.set_ray_reorder(True)
    gh.trace_status()
    assert torch.equal(hc, hc2) and torch.equal(cu.view(torch.int32), cu2.view(torch.int32))
    st = gh.trace_stats(rays, d, tree).cpu().numpy()
    assert np.array_equal(st[:, 3], hc.cpu().numpy())
    assert 5000 < hc.float().mean().item() < 8000                     # ~6.4e3 hits/ray
    sub = np.linspace(0, n_rays - 1, 512).astype(np.int64)
    rh = rays.cpu().numpy()[sub]; sh = d.cpu().numpy()
    assert np.array_equal(hc.cpu().numpy()[sub], oracle.brute_hitcounts(rh, sh))
    c32, c64 = oracle.brute_cumulative(rh, sh)
    check_column_densities(cu.cpu().numpy()[sub], c32, c64, integral_mode)


def test_config4_full_size_properties(gh, oracle, cuda, integral_mode):
    """project_gadget at 10^7 particles / 1024^2 rays through project_sph: image identical
    for the two packet orders, linear under ray subsetting (a shard traced alone gives the
    same pixels: the multi-GPU path), and exact against brute force on 48 pixels."""
    n, side = 10_000_000, 1024
    g = torch.Generator(device=cuda); g.manual_seed(42)
    s = torch.empty((n, 4), dtype=torch.float32, device=cuda)
    s[:, :3] = torch.rand((n, 3), generator=g, device=cuda)
    s[:, 3] = float((3.0 * 48.0 / (4.0 * np.pi * n)) ** (1.0 / 3.0))
    image, tree, rays = gh.project_sph(s, side, 32)
    gh.trace_status()
    img = image.reshape(-1)
    from grace_hip import sharding
    lo, hi = sharding.shard_bounds(len(rays), 8, 5)                 # what rank 5 of 8 would trace
    part = torch.empty(hi - lo, dtype=torch.float32, device=cuda)
    gh.trace_cumulative_sph(rays[lo:hi].contiguous(), s, tree, part)
    assert torch.equal(part.view(torch.int32), img[lo:hi].view(torch.int32))
    gh.set_ray_reorder(False)
    part2 = torch.empty_like(part)
    gh.trace_cumulative_sph(rays[lo:hi].contiguous(), s, tree, part2)
    gh.set_ray_reorder(True)
    assert torch.equal(part.view(torch.int32), part2.view(torch.int32))
    sub = np.linspace(0, len(rays) - 1, 48).astype(np.int64)
    c32, c64 = oracle.brute_cumulative(rays.cpu().numpy()[sub], s.cpu().numpy())
    check_column_densities(img.cpu().numpy()[sub], c32, c64, integral_mode)
    # mean column density of a unit box of n unit-mass particles viewed along z is n
    assert abs(float(img.double().mean()) / n - 1.0) < 0.01


def test_config3_integrate_gadget(gh, oracle, cuda, tmp_path, integral_mode):
    """BASELINE configs[2] (tests/integrate_gadget): a synthetic 128^3 Gadget-2 snapshot read
    from disk, one source at the box centre, HEALPix Nside 64 rays (49 152).  Column densities
    bit-equal to the oracle on a 384-ray subset, packet-order independent, plus the
    plane-parallel volume-integral KAT on the same snapshot."""
    from grace_hip import gadget
    n_side = 128
    n = n_side ** 3
    rng = np.random.default_rng(42)
    grid = np.stack(np.meshgrid(*[np.arange(n_side, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3)
    pos = ((grid + rng.random((n, 3), dtype=np.float32)) / n_side).astype(np.float32)
    h = np.full(n, (3 * 48 / (4 * np.pi * n)) ** (1 / 3), np.float32)
    fname = str(tmp_path / "Data_synth")
    gadget.write_gadget(fname, pos, h)
    s = gadget.read_gadget(fname)
    assert len(s) == 2097152
    d = _dev(s, cuda)
    lo, hi = gh.min_max_vec4(d)
    tree = gh.Tree(n, 32, device=cuda)
    gh.build_tree(d, tree, lo[:3], hi[:3])
    centre = (lo[:3] + hi[:3]) / 2
    length = float(np.linalg.norm(hi[:3] - lo[:3]))
    rays = gh.healpix_rays(64, centre, length, device=cuda)
    assert len(rays) == 49152
    out = torch.empty(len(rays), dtype=torch.float32, device=cuda)
    gh.trace_cumulative_sph(rays, d, tree, out)
    gh.set_ray_reorder(False)
    out2 = torch.empty_like(out)
    gh.trace_cumulative_sph(rays, d, tree, out2)
    gh.set_ray_reorder(True)
    gh.trace_status()
    assert torch.equal(out.view(torch.int32), out2.view(torch.int32))
    sub = np.linspace(0, len(rays) - 1, 384).astype(np.int64)
    c32, c64 = oracle.brute_cumulative(rays.cpu().numpy()[sub], d.cpu().numpy())
    check_column_densities(out.cpu().numpy()[sub], c32, c64, integral_mode)
    # integrate_gadget.cu:76-90 on the same snapshot (plane-parallel grid, w = max h)
    prays, area = gh.orthogonal_rays_z(512, lo, hi, device=cuda)
    pout = torch.empty(len(prays), dtype=torch.float32, device=cuda)
    gh.trace_cumulative_sph(prays, d, tree, pout)
    assert abs(1.0 - float(pout.double().sum()) * area / n) < 5e-4


def test_generic_functor_trace_program(tmp_path):
    """include/grace/hip/trace.hpp (the functor-parameterised kernel for custom primitives /
    payloads) built with hipcc: reference-style functor compositions == the built-in kernels
    (hit counts exact, sums bit for bit) and a user-defined functor == a host loop."""
    lib = os.path.join(ROOT, "grace-devel_amd", "lib")
    exe = tmp_path / "generic_trace"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17",
                           "-ffp-contract=off", "-w", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "generic_trace.hip"), "-o", str(exe),
                           "-L" + lib, "-lgrace_hip", "-L" + os.path.join(ROOT, "oracle"),
                           "-lgrace_oracle", "-Wl,-rpath," + lib,
                           "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PASSED" in r.stdout, r.stdout + r.stderr


def test_cpp_ray_generators_program(tmp_path):
    """tests/cpp/ray_generators.cpp: the gen_rays.cuh generators and the double4 key/sort
    overloads through include/grace/grace.h (plain g++)."""
    lib = os.path.join(ROOT, "grace-devel_amd", "lib")
    exe = tmp_path / "ray_generators"
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "ray_generators.cpp"), "-o", str(exe),
                           "-L" + lib, "-lgrace_hip", "-Wl,-rpath," + lib])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PASSED" in r.stdout, r.stdout + r.stderr


def _build_cpp(tmp_path, name):
    lib = os.path.join(ROOT, "grace-devel_amd", "lib")
    exe = tmp_path / name
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", str(exe),
                           "-L" + lib, "-lgrace_hip", "-Wl,-rpath," + lib])
    return str(exe)


def test_cpp_profile_tree_program(tmp_path):
    """tests/cpp/profile_tree.cpp prints the reference profile_tree's lines (one per build
    phase), scrapable the way tests/profile_leafbuilders.py scrapes them."""
    exe = _build_cpp(tmp_path, "profile_tree")
    r = subprocess.run([exe, "32", "3", "16", "17"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    assert "Max particles per leaf:   32" in out and "Iterations per tree:      3" in out
    blocks = out.split("Number of particles:")[1:]
    assert [int(b.split()[0]) for b in blocks] == [65536, 131072]
    for b in blocks:
        times = {}
        for label in ("Morton key generation", "sort-by-key", "computing deltas", "building leaves",
                      "computing leaf deltas", "building nodes", "total (inc. memory ops)"):
            line = [l for l in b.splitlines() if l.startswith("Time for " + label + ":")]
            assert len(line) == 1 and line[0].rstrip().endswith("ms.")
            times[label] = float(line[0].split(":")[1].split()[0])
        assert times["building leaves"] > 0 and times["building nodes"] > 0
        assert times["total (inc. memory ops)"] >= times["sort-by-key"] > 0


def test_cpp_profile_trace_gadget_program(tmp_path):
    """tests/cpp/profile_trace_gadget.cpp prints the reference profile_trace_gadget's lines."""
    exe = _build_cpp(tmp_path, "profile_trace_gadget")
    r = subprocess.run([exe, "64", "32", "synthetic:200000", "2"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    assert "Number of particles:    200000" in out and "Number of rays:         2048" in out
    hits = int([l for l in out.splitlines() if l.startswith("Total hits:")][0].split()[2])
    assert hits > 2048
    for label in ("generating and sorting rays", "hit count tracing", "cumulative density tracing",
                  "full tracing", "sort-by-distance", "total (inc. memory ops)"):
        line = [l for l in out.splitlines() if l.startswith("Time for " + label + ":")]
        assert len(line) == 1 and float(line[0].split(":")[1].split()[0]) > 0


def test_cpp_project_gadget_program(tmp_path):
    """tests/cpp/project_gadget.cpp (mirror of tests/project_gadget): reads a Gadget file
    written by the Python writer, projects it, writes a valid 24-bit BMP."""
    from grace_hip import gadget
    lib = os.path.join(ROOT, "grace-devel_amd", "lib")
    exe = tmp_path / "project_gadget"
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "project_gadget.cpp"), "-o", str(exe),
                           "-L" + lib, "-lgrace_hip", "-Wl,-rpath," + lib])
    rng = np.random.default_rng(8)
    n = 100000
    pos = rng.random((n, 3), dtype=np.float32)
    h = np.full(n, (3 * 48 / (4 * np.pi * n)) ** (1 / 3), np.float32)
    snap = str(tmp_path / "snap"); bmp = str(tmp_path / "density.bmp")
    gadget.write_gadget(snap, pos, h)
    r = subprocess.run([str(exe), "2048", "32", snap, bmp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Number of particles:     100000" in r.stdout and "Number of rays per side: 256" in r.stdout
    mean = float([l for l in r.stdout.splitlines() if l.startswith("Mean output")][0].split()[2])
    assert abs(mean / n - 1.0) < 0.05          # unit box, unit masses: mean column density = N
    raw = open(bmp, "rb").read()
    assert raw[:2] == b"BM" and len(raw) == 54 + 256 * 256 * 3
    assert int.from_bytes(raw[18:22], "little") == 256 and raw[28] == 24
    px = np.frombuffer(raw[54:], np.uint8)
    assert px.max() > 100 and px.min() < px.max()


# ---- double4 spheres (Real4 = double4, Real = double) --------------------------------------------
@pytest.mark.parametrize("n,mpl,n_rays", [(5000, 8, 256), (60000, 32, 1024)])
def test_double4_build_and_trace(gh, oracle, cuda, n, mpl, n_rays):
    """The reference's templates instantiated with double4 spheres (build_sph.cuh:84-126,
    trace_sph.cuh:57-110): keys from float-narrowed centres, Euclidean deltas formed in double and
    stored as float, the same ALBVH with float boxes of the double spheres, sphere_hit and the
    kernel integral in double.  Tree bit-identical to the oracle; hit counts == the double brute
    force; column densities bit-equal to the oracle's class-ordered double sum (the float path's
    summation order, in double: it lets a double4 packet be shared by several waves) and within
    1e-13 of the reference's single running double sum."""
    rng = np.random.default_rng(n)
    s = rng.uniform(0, 1, (n, 4)); s[:, 3] = rng.uniform(0.002, 0.05, n)
    d = _dev(s, cuda)
    tree = gh.Tree(n, mpl, device=cuda)
    bot, top = np.float32([0, 0, 0]), np.float32([1, 1, 1])
    gh.build_tree_d4(d, tree, bot, top)
    # oracle: same keys (narrowed centres), stable sort, double deltas, ALBVH with kind-2 boxes
    p4 = np.zeros((n, 4), np.float32); p4[:, :3] = s[:, :3].astype(np.float32)
    keys = oracle.morton_keys30(p4, bot, top)
    order = np.argsort(keys, kind="stable")
    ss = np.ascontiguousarray(s[order])
    assert np.array_equal(d.cpu().numpy(), ss)
    dl = oracle.deltas_euclid_d4(ss)
    nodes, leaves, root, _ = oracle.albvh(ss, dl, mpl, prim_kind=2)
    assert np.array_equal(tree.leaves.cpu().numpy(), leaves)
    assert np.array_equal(tree.nodes.cpu().numpy(), nodes)
    assert int(tree.root_index.item()) == root
    rays = gh.uniform_random_rays(n_rays, (0.5, 0.5, 0.5), 2.0, seed=8, device=cuda)
    hc = torch.empty(n_rays, dtype=torch.int32, device=cuda)
    cu = torch.empty(n_rays, dtype=torch.float64, device=cuda)
    gh.trace_hitcounts_d4(rays, d, tree, hc)
    gh.trace_cumulative_d4(rays, d, tree, cu)
    rr = rays.cpu().numpy()
    assert np.array_equal(hc.cpu().numpy(), oracle.brute_hitcounts_d4(rr, ss))
    ref = oracle.brute_cumulative_d4(rr, ss)
    got = cu.cpu().numpy()
    assert ref.sum() > 0 and np.array_equal(got == 0, ref == 0)
    assert np.array_equal(got, ref)       # same operations in the same order: the same doubles
    running = oracle.brute_cumulative_d4(rr, ss, blocks=1)     # the reference's order
    assert np.allclose(got, running, rtol=1e-13, atol=0)
