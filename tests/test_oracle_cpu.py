"""CPU tests (no GPU): the oracle against the reference's known-answer vectors and the
committed fixtures, plus the oracle's own internal consistency (tree walk == brute force)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def kat():
    return json.load(open(os.path.join(GOLD, "kat.json")))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "pipeline_n4096.npz"))


def test_morton_known_answers(oracle, kat):
    L = oracle.lib()
    k = kat["morton30"]   # tests/morton_key/30bit_key.cu:20-26
    assert L.go_space_by_two_10bit(k["x"]) == k["spaced_x"]
    assert L.go_space_by_two_10bit(k["y"]) == k["spaced_y"]
    assert L.go_space_by_two_10bit(k["z"]) == k["spaced_z"]
    assert L.go_morton_key30(k["x"], k["y"], k["z"]) == k["key"]
    k = kat["morton63"]   # tests/morton_key/63bit_key.cu:20-26
    assert L.go_space_by_two_21bit(k["x"]) == k["spaced_x"]
    assert L.go_space_by_two_21bit(k["y"]) == k["spaced_y"]
    assert L.go_space_by_two_21bit(k["z"]) == k["spaced_z"]
    assert L.go_morton_key63(k["x"], k["y"], k["z"]) == k["key"]


def test_morton_float_overloads(oracle):
    L = oracle.lib()
    # generic/morton.h:32-42: span * x truncated, then the integer overload
    assert L.go_morton_key30_unit(0.5, 0.25, 0.75) == L.go_morton_key30(511, 255, 767)
    assert L.go_morton_key63_unit(0.5, 0.25, 0.75) == L.go_morton_key63(1048575, 524287, 1572863)


def test_test_generator_known_answers(oracle, kat):
    k = kat["random_real4"]
    got = oracle.random_real4(3, k["low"], k["high"])
    assert np.array_equal(got, np.array(k["values"], np.float32))


def test_healpix_against_reference_run(oracle, kat):
    ref = np.load(os.path.join(GOLD, "healpix_nside4_ref.npy"))   # the reference's own output
    got = oracle.healpix_dirs(4)
    assert np.array_equal(got, ref)
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-15)
    v = oracle.healpix_dirs(64)[0]
    assert np.allclose(v, kat["healpix_nside64_pix0"], rtol=0, atol=1e-16)
    live = oracle.ref_healpix()      # only where /root/reference was present at build time
    if live is not None:
        out = (C.c_double * 3)()
        for nside in (1, 2, 8, 32):
            d = oracle.healpix_dirs(nside)
            for i in range(0, 12 * nside * nside, max(1, nside)):
                live.pix2vec_nest(nside, i, out)
                assert tuple(d[i]) == tuple(out)


def test_pipeline_fixture(oracle, gold):
    s = gold["spheres"]
    assert np.array_equal(oracle.random_real4(len(s), (0, 0, 0, 0), (1, 1, 1, 0.1)), s)
    assert np.array_equal(oracle.morton_keys30(s, (0, 0, 0), (1, 1, 1)), gold["keys30"])
    assert np.array_equal(oracle.morton_keys63(s, (0, 0, 0), (1, 1, 1)), gold["keys63"])
    keys, ss, order = oracle.sort_by_key(gold["keys30"], s)
    assert np.array_equal(order.astype(np.uint32), gold["order"])
    assert np.all(np.diff(keys.astype(np.int64)) >= 0)
    ss = np.ascontiguousarray(ss)
    d = oracle.deltas_euclid(ss)
    assert np.array_equal(d.view(np.uint32), gold["deltas"].view(np.uint32))
    for mpl in (1, 8, 32):
        nodes, leaves, root, _ = oracle.albvh(ss, d, mpl)
        assert np.array_equal(nodes, gold["nodes_%d" % mpl])
        assert np.array_equal(leaves, gold["leaves_%d" % mpl])
        assert root == int(gold["root_%d" % mpl])
    assert np.array_equal(oracle.brute_hitcounts(gold["rays"], ss), gold["hit_counts"])
    c32, c64 = oracle.brute_cumulative(gold["rays"], ss)
    assert np.array_equal(c32.view(np.uint32), gold["cumulative32"].view(np.uint32))
    assert np.array_equal(c64, gold["cumulative64"])
    assert np.array_equal(oracle.segscan(gold["seg_offsets"], gold["seg_data"]), gold["seg_result"])


@pytest.mark.parametrize("mpl", [1, 8, 32])
def test_tree_structure_invariants(oracle, gold, mpl):
    """What the reference's kernels promise: leaves partition the primitives in order, each
    holds <= max_per_leaf, every node's children cover adjacent leaf ranges and its child
    boxes contain their spheres."""
    nodes = gold["nodes_%d" % mpl]; leaves = gold["leaves_%d" % mpl]; root = int(gold["root_%d" % mpl])
    n_nodes = len(nodes)
    assert leaves[0, 0] == 0 and leaves[:, 1].sum() == 4096
    assert np.all(leaves[1:, 0] == leaves[:-1, 0] + leaves[:-1, 1])
    assert leaves[:, 1].max() <= mpl and leaves[:, 1].min() >= 1
    assert nodes[root, 2] == 0 and nodes[root, 3] == len(leaves) - 1
    seen = np.zeros(n_nodes + len(leaves), bool)
    for i in range(n_nodes):
        l, r, first, last = nodes[i, :4]
        assert not seen[l] and not seen[r]
        seen[l] = seen[r] = True
        lr = (nodes[l, 2], nodes[l, 3]) if l < n_nodes else (l - n_nodes, l - n_nodes)
        rr = (nodes[r, 2], nodes[r, 3]) if r < n_nodes else (r - n_nodes, r - n_nodes)
        assert lr[0] == first and rr[1] == last and lr[1] + 1 == rr[0] == i + 1
    assert seen.sum() == n_nodes + len(leaves) - 1 and not seen[root]


@pytest.mark.parametrize("width", [1, 32, 64])
def test_tree_walk_equals_brute_force(oracle, gold, width):
    """tests/tree_traversal/tree_traversal.cu:65-100 inside the oracle itself, for the
    reference's 32-wide packets, 64-wide packets and single rays; per-ray statistics must
    not depend on the packet width."""
    s = gold["spheres"]
    _, ss, _ = oracle.sort_by_key(gold["keys30"], s)
    ss = np.ascontiguousarray(ss)
    nodes = gold["nodes_32"].view(np.float32); leaves = gold["leaves_32"]; root = int(gold["root_32"])
    hc, st = oracle.trace(gold["rays"], ss, nodes, leaves, root, width=width, mode=0, stats=True)
    assert np.array_equal(hc, gold["hit_counts"])
    _, st1 = oracle.trace(gold["rays"], ss, nodes, leaves, root, width=1, mode=0, stats=True)
    assert np.array_equal(st, st1)
    cu = oracle.trace(gold["rays"], ss, nodes, leaves, root, width=width, mode=1)
    assert np.array_equal(cu.view(np.uint32), gold["cumulative32"].view(np.uint32))


def test_volume_integral_kat(oracle, kat):
    """tests/integrate/integrate.cu:21-43,79-101: two spheres of radius 0.2 at (-+0.5)^3 in
    [-1,1]^3; sum of ray integrals x area per ray / N == 1 +- 5e-4 (each SPH kernel
    integrates to one).  Pins sphere_hit + lerp + the table end to end."""
    s = np.array([[-0.5, -0.5, -0.5, 0.2], [0.5, 0.5, 0.5, 0.2]], np.float32)
    rays, area = oracle.orthogonal_rays_z(512, (-1, -1, -1, 0.2), (1, 1, 1, 0.2))
    c32, c64 = oracle.brute_cumulative(rays, s)
    integral = np.float32(c32.sum(dtype=np.float32) * np.float32(area)) / np.float32(2)
    assert abs(1.0 - float(integral)) < kat["integrate_tolerance"]
    assert abs(1.0 - c64.sum() * area / 2) < kat["integrate_tolerance"]


def test_sphere_hit_edge_cases(oracle):
    """generic/intersect.h:37-54: origin inside beyond closest approach and terminus short of
    closest approach are misses; partial intersections around the closest approach hit."""
    L = oracle.lib()
    L.go_sphere_hit.restype = C.c_int
    def hit(ray, sph):
        r = np.array([ray], np.float32); s = np.array([sph], np.float32)
        b2 = C.c_float(); d = C.c_float()
        return L.go_sphere_hit(C.c_void_p(r.ctypes.data), C.c_void_p(s.ctypes.data), C.byref(b2), C.byref(d)), b2.value, d.value
    assert hit([1, 0, 0, 0, 0, 0, 10], [5, 0, 0, 1])[0] == 1
    assert hit([1, 0, 0, 0, 0, 0, 10], [5, 0.999, 0, 1])[0] == 1
    assert hit([1, 0, 0, 0, 0, 0, 10], [5, 1.0, 0, 1])[0] == 0          # b2 >= r^2
    assert hit([1, 0, 0, 0, 0, 0, 10], [-0.5, 0, 0, 1])[0] == 0         # dot_p < 0
    assert hit([1, 0, 0, 0, 0, 0, 5], [5, 0, 0, 1])[0] == 0             # dot_p >= length
    assert hit([1, 0, 0, 0, 0, 0, 5.5], [5, 0, 0, 1])[0] == 1           # ends inside, past centre
    assert hit([1, 0, 0, 0, 0, 0, 10], [0.5, 0, 0, 1])[0] == 1          # starts inside, before centre


def test_slab_test_matches_float_semantics(oracle, gold):
    """The integer min/max trick (device/intrinsics.cuh) agrees with a plain float slab test
    (tests/AABB_intersect/williams.cu:40-61 form) on finite boxes."""
    nodes = gold["nodes_8"].view(np.float32)
    rays = gold["rays"]
    rng = np.random.default_rng(1)
    for ni in rng.integers(0, len(nodes), 200):
        nf = nodes[ni]
        for ri in rng.integers(0, len(rays), 5):
            r = rays[ri]
            got = oracle.aabbs_hit(r[None, :], nf)
            inv = 1.0 / r[:3].astype(np.float32)
            o = r[3:6]
            def slab(b):  # b = (bx,tx,by,ty,bz,tz)
                t0 = (np.float32(b[0::2]) - o) * inv; t1 = (np.float32(b[1::2]) - o) * inv
                tmin = max(np.minimum(t0, t1).max(), np.float32(0)); tmax = min(np.maximum(t0, t1).min(), r[6])
                return tmax >= tmin
            L = (nf[4], nf[5], nf[6], nf[7], nf[12], nf[13]); R = (nf[8], nf[9], nf[10], nf[11], nf[14], nf[15])
            assert got == int(slab(R)) + 2 * int(slab(L))


def test_albvh_rejects_small_input(oracle):
    s = oracle.random_real4(8, (0, 0, 0, 0), (1, 1, 1, 0.1))
    with pytest.raises(ValueError):      # albvh.cuh:795-799
        oracle.albvh(s, oracle.deltas_euclid(s), 8)


def test_segscan_semantics(oracle):
    offs = np.array([0, 3, 3, 7], np.int32)          # the example of trace_sph.cuh:130-134
    data = np.arange(1, 9, dtype=np.float32)
    assert oracle.segscan(offs, data).tolist() == [0, 1, 3, 0, 4, 9, 15, 0]
    out, total = oracle.exclusive_scan_i32(np.array([3, 0, 4, 1], np.int32))
    assert out.tolist() == [0, 3, 3, 7] and total == 8


def test_block_ordered_sum_vs_single_running_sum(oracle, gold):
    """The stated result (8 interleaved primitive classes, summed pairwise) against the reference's
    single fp32 running sum (blocks=1) and the fp64 sum: all within 1e-6."""
    s = gold["spheres"]
    _, ss, _ = oracle.sort_by_key(gold["keys30"], s)
    ss = np.ascontiguousarray(ss)
    c16, c64 = oracle.brute_cumulative(gold["rays"], ss)
    c1, _ = oracle.brute_cumulative(gold["rays"], ss, blocks=1)
    nz = c64 > 0
    assert np.all(np.abs(c16[nz] - c64[nz]) <= 1e-6 * c64[nz])
    assert np.all(np.abs(c1[nz] - c64[nz]) <= 1e-6 * c64[nz])
    assert np.all(np.abs(c16[nz] - c1[nz]) <= 4e-7 * c64[nz])
