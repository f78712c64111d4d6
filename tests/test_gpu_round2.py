"""GPU parity, round 2: the reference-order running sum, the prepared scene, cluster culling
against brute force on clustered scenes, the remaining type instantiations (64-bit XOR deltas,
double deltas, double4 area deltas / per-hit trace / sort_by_distance), the INT32_MAX guard of the
per-hit offsets, zero-ray calls, and a real two-rank sharded trace gathered under gloo."""
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def _gadget_like(n, cuda, seed=42):
    g = torch.Generator(device=cuda); g.manual_seed(seed)
    s = torch.empty((n, 4), dtype=torch.float32, device=cuda)
    s[:, :3] = torch.rand((n, 3), generator=g, device=cuda)
    s[:, 3] = float((3.0 * 48.0 / (4.0 * math.pi * n)) ** (1.0 / 3.0))
    return s


# ---- the REFERENCE's summation order ----------------------------------------
@pytest.mark.parametrize("config", ["config2", "config3", "config4"])
def test_column_densities_against_reference_order_running_sum(gh, oracle, cuda, config):
    """The reference keeps ONE running fp32 sum per ray in ascending primitive order
    (functors/trace.cuh:191); this implementation's stated value is the class-ordered sum.
    Both evaluations (default and exact) are compared with the oracle's blocks = 1 sum -- the
    reference's actual definition -- on BASELINE's configurations, under north_star's stated fp32
    tolerance 1e-5.  A running fp32 sum of H terms is itself only good to ~sqrt(H) ulp: on
    config 2 (6400 hits per ray) the REFERENCE-order sum sits up to ~5e-6 from the fp64 sum, so
    no tighter bound against it is meaningful there; what is asserted besides the tolerance is
    that this implementation is never further from the fp64 sum than the reference's order is
    (plus 1e-6).  Measured maxima are printed (pytest -s) and recorded in DESIGN.md."""
    if config == "config2":       # 10^6 random spheres r in U[0, 0.1), isotropic rays from the centre
        n = 1_000_000
        s = _dev(oracle.random_real4(n, (0, 0, 0, 0), (1, 1, 1, 0.1)), cuda)
        lo, hi = (0, 0, 0), (1, 1, 1)
        rays = gh.uniform_random_rays(100_000, (0.5, 0.5, 0.5), 2.0, seed=1234, device=cuda)
    elif config == "config3":     # 128^3 jittered lattice, HEALPix Nside 64 from the centre
        ns = 128; n = ns ** 3
        g = torch.Generator(device=cuda); g.manual_seed(42)
        grid = torch.stack(torch.meshgrid(*[torch.arange(ns, device=cuda)] * 3, indexing="ij"), -1).reshape(-1, 3).float()
        pos = (grid + torch.rand((n, 3), generator=g, device=cuda)) / ns
        s = torch.cat([pos, torch.full((n, 1), (3 * 48 / (4 * math.pi * n)) ** (1 / 3), device=cuda)], 1).contiguous()
        lo4, hi4 = gh.min_max_vec4(s); lo, hi = lo4[:3], hi4[:3]
        rays = gh.healpix_rays(64, (lo4[:3] + hi4[:3]) / 2, float(np.linalg.norm(hi4[:3] - lo4[:3])), device=cuda)
    else:                         # 10^7 particles, 1024^2 orthographic rays
        n = 10_000_000
        s = _gadget_like(n, cuda)
        lo4, hi4 = gh.min_max_vec4(s); lo4[3] = hi4[3] = 0; lo, hi = lo4[:3], hi4[:3]
        rays, _ = gh.orthogonal_rays_z(1024, lo4, hi4, device=cuda)
    tree = gh.Tree(n, 32, device=cuda)
    gh.build_tree(s, tree, lo, hi)
    out = torch.empty(len(rays), dtype=torch.float32, device=cuda)
    sub = np.linspace(0, len(rays) - 1, 96 if config == "config4" else 192).astype(np.int64)
    ref_run, ref64 = oracle.brute_cumulative(rays.cpu().numpy()[sub], s.cpu().numpy(), blocks=1)
    assert ref_run.min() > 0
    for exact in (False, True):
        gh.set_exact_integrals(exact)
        try:
            gh.trace_cumulative_sph(rays, s, tree, out, check=True)
        finally:
            gh.set_exact_integrals(False)
        got = out.cpu().numpy()[sub].astype(np.float64)
        rel = np.abs(got - ref_run.astype(np.float64)) / ref_run
        ours64 = np.abs(got - ref64) / ref64
        ref64_err = np.abs(ref_run.astype(np.float64) - ref64) / ref64
        print("%s exact=%d: max |ours - reference-order sum| %.2e; |ours - fp64| %.2e; "
              "|reference-order - fp64| %.2e" % (config, exact, rel.max(), ours64.max(), ref64_err.max()))
        assert rel.max() < 1e-5, (config, exact, rel.max())
        assert ours64.max() <= ref64_err.max() + 1e-6, (config, exact, ours64.max(), ref64_err.max())


# ---- prepared scene ---------------------------------------------------------------------------
def test_prepared_scene_is_bit_identical_and_invalidated_by_library_writes(gh, oracle, cuda):
    n = 200_000
    s = _dev(oracle.random_real4(n, (0, 0, 0, 0.002), (1, 1, 1, 0.02)), cuda)
    tree = gh.Tree(n, 32, device=cuda)
    gh.build_tree(s, tree, (0, 0, 0), (1, 1, 1))
    rays = gh.uniform_random_rays(32 * 200, (0.5, 0.5, 0.5), 2.0, seed=5, device=cuda)
    lo4, hi4 = gh.min_max_vec4(s)
    orays, _ = gh.orthogonal_rays_z(96, lo4, hi4, device=cuda)
    ref = {}
    for name, r in (("iso", rays), ("ortho", orays)):
        hc = torch.empty(len(r), dtype=torch.int32, device=cuda); cu = torch.empty(len(r), dtype=torch.float32, device=cuda)
        gh.trace_hitcounts_sph(r, s, tree, hc); gh.trace_cumulative_sph(r, s, tree, cu)
        ref[name] = (hc.clone(), cu.clone(), [t.clone() for t in gh.trace_sph(r, s, tree)])
    gh.trace_prepare(s, tree)
    try:
        for name, r in (("iso", rays), ("ortho", orays)):
            hc = torch.empty(len(r), dtype=torch.int32, device=cuda); cu = torch.empty(len(r), dtype=torch.float32, device=cuda)
            gh.trace_hitcounts_sph(r, s, tree, hc); gh.trace_cumulative_sph(r, s, tree, cu, check=True)
            assert torch.equal(hc, ref[name][0]) and torch.equal(cu.view(torch.int32), ref[name][1].view(torch.int32))
            for a, b in zip(gh.trace_sph(r, s, tree), ref[name][2]):
                assert torch.equal(a, b)
            gh.set_exact_integrals(True)
            cu2 = torch.empty_like(cu); gh.trace_cumulative_sph(r, s, tree, cu2)
            gh.set_exact_integrals(False)
            c32, c64 = oracle.brute_cumulative(r.cpu().numpy()[:64], s.cpu().numpy())
            assert np.array_equal(cu2.cpu().numpy()[:64].view(np.uint32), c32.view(np.uint32))
        # the library's own writes to a prepared array drop the cache: scale the radii, re-sort
        # (identity permutation, but a write) -- the next trace must see the new radii
        s[:, 3] *= 0.5
        keys = torch.empty(n, dtype=torch.int32, device=cuda)
        gh.morton_keys_sph(s, keys, (0, 0, 0), (1, 1, 1)); gh.sort_by_key(keys, s, 0, 30)
        hc = torch.empty(len(rays), dtype=torch.int32, device=cuda)
        gh.trace_hitcounts_sph(rays, s, tree, hc, check=True)
        assert np.array_equal(hc.cpu().numpy()[:128], oracle.brute_hitcounts(rays.cpu().numpy()[:128], s.cpu().numpy()))
    finally:
        gh.trace_release()


# ---- prepared ray batch -------------------------------------------------------------------------
def test_prepared_rays_are_bit_identical_and_dropped_when_a_generator_rewrites_them(gh, oracle, cuda):
    import ctypes as C
    n = 150_000
    s = _dev(oracle.random_real4(n, (0, 0, 0, 0.002), (1, 1, 1, 0.03)), cuda)
    tree = gh.Tree(n, 32, device=cuda)
    gh.build_tree(s, tree, (0, 0, 0), (1, 1, 1))
    lo4, hi4 = gh.min_max_vec4(s)
    batches = {"iso": gh.uniform_random_rays(32 * 150, (0.5, 0.5, 0.5), 2.0, seed=5, device=cuda),
               "ortho": gh.orthogonal_rays_z(88, lo4, hi4, device=cuda)[0],
               "healpix": gh.healpix_rays(16, (0.4, 0.5, 0.6), 2.0, device=cuda)}
    def outputs(r):
        hc = torch.empty(len(r), dtype=torch.int32, device=cuda); cu = torch.empty(len(r), dtype=torch.float32, device=cuda)
        gh.trace_hitcounts_sph(r, s, tree, hc, check=True); gh.trace_cumulative_sph(r, s, tree, cu, check=True)
        return [hc, cu.view(torch.int32)] + [t.view(torch.int32) for t in gh.trace_sph(r, s, tree)]
    try:
        for name, r in batches.items():
            ref = [t.clone() for t in outputs(r)]
            gh.trace_prepare_rays(r)
            for _ in range(2):
                for a, b in zip(outputs(r), ref):
                    assert torch.equal(a, b), name
            other = batches["iso" if name != "iso" else "ortho"]          # another batch: cache not used
            hc = torch.empty(len(other), dtype=torch.int32, device=cuda)
            gh.trace_hitcounts_sph(other, s, tree, hc, check=True)
            assert np.array_equal(hc.cpu().numpy()[:64], oracle.brute_hitcounts(other.cpu().numpy()[:64], s.cpu().numpy()))
        # a library ray generator writing into the prepared array drops the cache: the next trace
        # must order the NEW rays (a stale order would still give right results -- packets are only
        # a grouping -- so check through the timing-free route: prepare, rewrite, trace == brute force
        # and the cache pointer no longer matches: a second prepare of the same tensor succeeds)
        r = batches["healpix"]
        gh.trace_prepare_rays(r)
        gh._check(gh._lib.grace_rays_healpix(C.c_int(16), C.c_float(0.7), C.c_float(0.3), C.c_float(0.5),
                                             C.c_float(1.5), gh._ptr(r), gh._stream()))
        hc = torch.empty(len(r), dtype=torch.int32, device=cuda)
        gh.trace_hitcounts_sph(r, s, tree, hc, check=True)
        sub = slice(0, len(r), 13)
        assert np.array_equal(hc.cpu().numpy()[sub], oracle.brute_hitcounts(r.cpu().numpy()[sub], s.cpu().numpy()))
    finally:
        gh.trace_release_rays()


# ---- cluster culling on a strongly clustered scene ---------------------------------------------
@pytest.mark.parametrize("kind", ["ortho", "iso", "plane"])
def test_cluster_culling_on_clustered_scene_equals_brute_force(gh, oracle, cuda, kind):
    """Clumps of very different density and radius (cluster boxes of very different sizes, many
    duplicates of one position) with the large treelets the cluster tests are for."""
    rng = np.random.default_rng(8)
    centres = rng.random((40, 3)).astype(np.float32)
    which = rng.integers(0, 40, 120_000)
    scale = (10.0 ** rng.uniform(-3.5, -1.0, 40)).astype(np.float32)
    pos = centres[which] + rng.normal(size=(120_000, 3)).astype(np.float32) * scale[which][:, None]
    pos[:500] = pos[0]                                               # duplicates
    h = (scale[which] * rng.uniform(0.2, 3.0, 120_000)).astype(np.float32)
    s = np.concatenate([pos, h[:, None]], 1).astype(np.float32)
    lo = s[:, :3].min(0); hi = s[:, :3].max(0)
    d = _dev(s, cuda); tree = gh.Tree(len(s), 32, device=cuda)
    gh.build_tree(d, tree, lo, hi)
    if kind == "ortho":
        lo4 = np.append(lo, 0).astype(np.float32); hi4 = np.append(hi, float(h.max())).astype(np.float32)
        rays, _ = gh.orthogonal_rays_z(64, lo4, hi4, device=cuda)
    elif kind == "iso":
        rays = gh.uniform_random_rays(32 * 96, centres[3], 3.0, seed=9, device=cuda)
    else:
        rays = gh.plane_parallel_random_rays(64, 48, (lo[0], lo[1], lo[2] - 0.1), (hi[0] - lo[0], 0.1, 0),
                                             (0, hi[1] - lo[1], 0.05), 3.0, seed=3, device=cuda)
    sh = d.cpu().numpy(); rh = rays.cpu().numpy()
    ref = oracle.brute_hitcounts(rh, sh)
    assert ref.sum() > 0
    for T in (-1, 4096, 65536, 0):
        gh.set_treelet_size(T)
        try:
            hc = torch.empty(len(rays), dtype=torch.int32, device=cuda)
            gh.trace_hitcounts_sph(rays, d, tree, hc, check=True)
            assert np.array_equal(hc.cpu().numpy(), ref), (kind, T)
            offs, idx, w, dist = gh.trace_sph(rays, d, tree)
            o2, i2, w2, d2 = oracle.brute_hits(rh, sh)
            assert np.array_equal(offs.cpu().numpy(), o2) and np.array_equal(idx.cpu().numpy(), i2)
            assert np.array_equal(w.cpu().numpy().view(np.uint32), w2.view(np.uint32))
        finally:
            gh.set_treelet_size(-1)


# ---- origin-lattice cull: spheres smaller than the pixel spacing --------------------------------
@pytest.mark.parametrize("axis,sign", [(0, 1.0), (1, -1.0), (2, 1.0), (2, -1.0)])
def test_sub_pixel_spheres_between_the_rays_of_a_pixel_grid(gh, oracle, cuda, axis, sign):
    """Dense clumps whose spheres are smaller than the ray spacing (most fall between the rays,
    some sit exactly ON a ray, some are large): the lattice instantiation of the trace kernels
    (chosen on the device from the ray spacing and the scene's smallest sphere) must still equal
    brute force -- hit counts, per-hit outputs and the bit-exact column densities -- for pixel
    grids along +-x, +-y, +-z, for a grid that is NOT a multiple of 8 wide, with the caller's ray
    order (packets are then 64 x 1 strips: no lattice, same results) and with a prepared scene."""
    rng = np.random.default_rng(100 + axis)
    n = 90_000
    centres = rng.random((12, 3)).astype(np.float32) * 0.8 + 0.1
    which = rng.integers(0, 12, n)
    sig = (10.0 ** rng.uniform(-2.6, -1.3, 12)).astype(np.float32)
    pos = centres[which] + rng.normal(size=(n, 3)).astype(np.float32) * sig[which][:, None]
    side = 120                                                     # 120 x 120 rays: 15 x 15 tiles
    pitch = 1.0 / side
    h = (pitch * 10.0 ** rng.uniform(-1.7, 0.5, n)).astype(np.float32)   # 0.02 ... 3 pixels
    a1, a2 = [k for k in range(3) if k != axis]
    g1 = ((np.arange(side) + 0.5) * pitch).astype(np.float32)
    # a few hundred tiny spheres centred exactly on a ray, and a few exactly half-way between two
    on = rng.integers(0, side, (300, 2))
    pos[:300, a1] = g1[on[:, 0]]; pos[:300, a2] = g1[on[:, 1]]
    pos[300:400, a1] = (g1[on[:100, 0]] + np.float32(0.5 * pitch)).astype(np.float32)
    s = np.concatenate([pos, h[:, None]], 1).astype(np.float32)
    lo = s[:, :3].min(0) - 0.01; hi = s[:, :3].max(0) + 0.01
    d = _dev(s, cuda); tree = gh.Tree(n, 32, device=cuda)
    gh.build_tree(d, tree, lo, hi)
    sh = d.cpu().numpy()
    r = np.zeros((side * side, 7), dtype=np.float32)
    r[:, axis] = sign
    r[:, 3 + a1] = np.repeat(g1, side); r[:, 3 + a2] = np.tile(g1, side)
    r[:, 3 + axis] = lo[axis] - 0.05 if sign > 0 else hi[axis] + 0.05
    r[:, 6] = (hi[axis] - lo[axis]) + 0.1
    rays = _dev(r, cuda)
    ref = oracle.brute_hitcounts(r, sh)
    assert ref.sum() > 10_000 and (ref == 0).sum() > 0
    o2, i2, w2, d2 = oracle.brute_hits(r, sh)
    sub = slice(0, len(r), 7)
    c32, _ = oracle.brute_cumulative(r[sub], sh)
    for reorder, prepared in ((True, False), (True, True), (False, False)):
        gh.set_ray_reorder(reorder)
        if prepared:
            gh.trace_prepare(d, tree)
        try:
            hc = torch.empty(len(r), dtype=torch.int32, device=cuda)
            gh.trace_hitcounts_sph(rays, d, tree, hc, check=True)
            assert gh.last_lattice() == (1 if reorder else 0)      # the instantiation under test ran
            assert np.array_equal(hc.cpu().numpy(), ref), (axis, sign, reorder, prepared)
            offs, idx, w, dist = gh.trace_sph(rays, d, tree)
            assert np.array_equal(offs.cpu().numpy(), o2) and np.array_equal(idx.cpu().numpy(), i2)
            assert np.array_equal(w.cpu().numpy().view(np.uint32), w2.view(np.uint32))
            assert np.array_equal(dist.cpu().numpy().view(np.uint32), d2.view(np.uint32))
            gh.set_exact_integrals(True)
            cu = torch.empty(len(r), dtype=torch.float32, device=cuda)
            gh.trace_cumulative_sph(rays, d, tree, cu, check=True)
            gh.set_exact_integrals(False)
            assert np.array_equal(cu.cpu().numpy()[sub].view(np.uint32), c32.view(np.uint32))
            cf = torch.empty(len(r), dtype=torch.float32, device=cuda)
            gh.trace_cumulative_sph(rays, d, tree, cf, check=True)
            exact = cu.cpu().numpy().astype(np.float64); fast = cf.cpu().numpy().astype(np.float64)
            nz = exact > 0
            # north_star's fp32 tolerance on rays with enough hits for a relative bound to mean
            # something; a ray whose sum is one or two GRAZING hits (b -> h, F -> 0) is conditioned
            # by the fp32 rounding of b^2 itself -- fused in the fast path, unfused in the
            # reference's -- whatever evaluates the table: bounded by 1e-3 there
            rel = np.abs(fast - exact) / np.where(nz, exact, 1.0)
            many = ref >= 50
            assert rel[many].max() <= 1e-5 and rel[nz].max() <= 1e-3 and np.all(fast[~nz] == 0), \
                (rel[many].max(), rel[nz].max())
        finally:
            gh.set_ray_reorder(True)
            if prepared:
                gh.trace_release()


def test_clustered_scene_full_frame_splits_packets_and_stays_bit_identical(gh, oracle, cuda):
    """1024^2 orthographic rays (16384 packets: normally one wave each) through a clustered scene
    with sub-pixel spheres: the device flag sends the batch to the lattice instantiation with four
    waves per packet.  Hit counts and bit-exact column densities must equal the one-wave-per-packet
    trace (grace_trace_set_packet_split(1)) bit for bit, and brute force on a subset of the rays."""
    g = torch.Generator(device=cuda); g.manual_seed(5)
    n, nb = 1_200_000, 300_000
    pos = torch.rand((nb, 3), generator=g, device=cuda)
    dens = torch.full((nb,), float(nb), device=cuda)
    n_clumps = 20; nc = n - nb
    centres = torch.rand((n_clumps, 3), generator=g, device=cuda) * 0.8 + 0.1
    sig = 10 ** (torch.rand(n_clumps, generator=g, device=cuda) * 1.2 - 2.8)
    which = torch.randint(0, n_clumps, (nc,), generator=g, device=cuda)
    p = centres[which] + torch.randn((nc, 3), generator=g, device=cuda) * sig[which, None]
    r2 = ((p - centres[which]) ** 2).sum(1) / sig[which] ** 2
    d = (nc / n_clumps) * torch.exp(-0.5 * r2) / ((2 * math.pi) ** 1.5 * sig[which] ** 3) + nb
    pos = torch.cat([pos, p.clamp(0, 1)]); dens = torch.cat([dens, d])
    h = (3 * 48 / (4 * math.pi * dens)) ** (1 / 3)
    s = torch.cat([pos, h[:, None]], 1).float().contiguous()
    assert float(h.min()) < 0.5 / 1024                            # below the pixel spacing
    lo, hi = gh.min_max_vec4(s); lo[3] = hi[3] = 0
    tree = gh.Tree(n, 32, device=cuda); gh.build_tree(s, tree, lo[:3], hi[:3])
    rays, _ = gh.orthogonal_rays_z(1024, lo, hi, device=cuda)
    R = len(rays)
    out = {}
    for K in (-1, 1):
        gh.set_packet_split(K)
        try:
            hc = torch.empty(R, dtype=torch.int32, device=cuda); cu = torch.empty(R, dtype=torch.float32, device=cuda)
            gh.trace_hitcounts_sph(rays, s, tree, hc, check=True)
            assert gh.last_lattice() == 1
            gh.set_exact_integrals(True)
            gh.trace_cumulative_sph(rays, s, tree, cu, check=True)
            gh.set_exact_integrals(False)
            cf = torch.empty(R, dtype=torch.float32, device=cuda)
            gh.trace_cumulative_sph(rays, s, tree, cf, check=True)
            out[K] = (hc, cu, cf)
        finally:
            gh.set_packet_split(-1); gh.set_exact_integrals(False)
    for a, b in zip(out[-1], out[1]):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    sub = torch.arange(0, R, R // 192, device=cuda)[:192]
    rh = rays[sub].cpu().numpy(); sh = s.cpu().numpy()
    assert np.array_equal(out[-1][0][sub].cpu().numpy(), oracle.brute_hitcounts(rh, sh))
    c32, _ = oracle.brute_cumulative(rh, sh)
    assert np.array_equal(out[-1][1][sub].cpu().numpy().view(np.uint32), c32.view(np.uint32))
    # the same for a big one-origin batch (HEALPix Nside 512: 49152 packets): no lattice to cull
    # against, but the flag (ray spacing at the far end of the longest ray) splits its packets too
    prays = gh.healpix_rays(512, (0.45, 0.5, 0.55), 1.5, device=cuda)
    P = len(prays)
    pout = {}
    for K in (-1, 1):
        gh.set_packet_split(K)
        try:
            hc = torch.empty(P, dtype=torch.int32, device=cuda); cu = torch.empty(P, dtype=torch.float32, device=cuda)
            gh.trace_hitcounts_sph(prays, s, tree, hc, check=True)
            assert gh.last_lattice() == 1
            gh.set_exact_integrals(True)
            gh.trace_cumulative_sph(prays, s, tree, cu, check=True)
            gh.set_exact_integrals(False)
            pout[K] = (hc, cu)
        finally:
            gh.set_packet_split(-1); gh.set_exact_integrals(False)
    for a, b in zip(pout[-1], pout[1]):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    psub = torch.arange(0, P, P // 128, device=cuda)[:128]
    ph = prays[psub].cpu().numpy()
    assert np.array_equal(pout[-1][0][psub].cpu().numpy(), oracle.brute_hitcounts(ph, sh))
    pc32, _ = oracle.brute_cumulative(ph, sh)
    assert np.array_equal(pout[-1][1][psub].cpu().numpy().view(np.uint32), pc32.view(np.uint32))


# ---- remaining instantiations -----------------------------------------------------------------
def test_albvh_from_64bit_xor_deltas_and_double_deltas(gh, oracle, cuda):
    """morton_keys63_sort_sph -> XOR_deltas_sph<uinteger64> -> ALBVH_sph<float4, uinteger64>
    (build_sph.cuh:65-82, 108-124) == the oracle's tree; ALBVH from double-typed Euclidean deltas
    == the tree from the float ones."""
    n = 60_000
    s = oracle.random_real4(n, (0, 0, 0, 0.001), (1, 1, 1, 0.02))
    d = _dev(s, cuda)
    keys = gh.morton_keys63_sort_sph(d, (0, 0, 0), (1, 1, 1))
    dl = torch.empty(n + 1, dtype=torch.int64, device=cuda)
    gh.XOR_deltas_sph(keys, dl)
    ss = d.cpu().numpy()
    k_ref = oracle.morton_keys63(ss, (0, 0, 0), (1, 1, 1))
    assert np.array_equal(keys.cpu().numpy().view(np.uint64), k_ref)
    dl_ref = oracle.deltas_xor(k_ref)
    assert np.array_equal(dl.cpu().numpy().view(np.uint64), dl_ref)
    for mpl in (1, 16, 32):
        tree = gh.Tree(n, mpl, device=cuda)
        gh.ALBVH_sph(d, dl, tree)
        nodes, leaves, root, _ = oracle.albvh(ss, dl_ref, mpl)
        assert np.array_equal(tree.leaves.cpu().numpy(), leaves)
        assert np.array_equal(tree.nodes.cpu().numpy(), nodes) and int(tree.root_index.item()) == root
    df = torch.empty(n + 1, dtype=torch.float32, device=cuda)
    gh.euclidean_deltas_sph(d, df)
    t32 = gh.Tree(n, 32, device=cuda); gh.ALBVH_sph(d, df, t32)
    t64 = gh.Tree(n, 32, device=cuda); gh.ALBVH_sph(d, df.double(), t64)
    assert torch.equal(t32.nodes, t64.nodes) and torch.equal(t32.leaves, t64.leaves)


def test_double4_area_deltas_per_hit_trace_and_distance_sort(gh, oracle, cuda):
    n, n_rays = 50_000, 32 * 24
    s32 = oracle.random_real4(n, (0, 0, 0, 0.003), (1, 1, 1, 0.03))
    rng = np.random.default_rng(4)
    s = s32.astype(np.float64) + rng.uniform(-1e-9, 1e-9, s32.shape)      # genuinely double
    d = _dev(s, cuda)
    tree = gh.Tree(n, 32, device=cuda)
    gh.build_tree_d4(d, tree, (0, 0, 0), (1, 1, 1))
    ss = d.cpu().numpy()
    # surface_area_deltas_sph<double4>: AABBSphere narrows centre -+ radius to float3, area in fp32
    for dt in (torch.float32, torch.float64):
        da = torch.empty(n + 1, dtype=dt, device=cuda)
        gh.surface_area_deltas_d4(d, da)
        bot = (ss[:, :3] - ss[:, 3:4]).astype(np.float32); top = (ss[:, :3] + ss[:, 3:4]).astype(np.float32)
        L = np.maximum(top[:-1], top[1:]) - np.minimum(bot[:-1], bot[1:])
        sa = (L[:, 0] * L[:, 1]) + (L[:, 0] * L[:, 2]) + (L[:, 1] * L[:, 2])
        got = da.cpu().numpy()
        assert np.isinf(got[0]) and np.isinf(got[-1]) and np.array_equal(got[1:-1].astype(np.float32), sa)
    rays = gh.uniform_random_rays(n_rays, (0.5, 0.5, 0.5), 2.0, seed=21, device=cuda)
    rr = rays.cpu().numpy()
    offs, idx, w, dist = gh.trace_sph_d4(rays, d, tree)
    counts = oracle.brute_hitcounts_d4(rr, ss)
    assert np.array_equal(np.diff(np.append(offs.cpu().numpy(), len(idx))), counts)
    cu = torch.empty(n_rays, dtype=torch.float64, device=cuda)
    gh.trace_cumulative_d4(rays, d, tree, cu)
    assert np.array_equal(cu.cpu().numpy(), oracle.brute_cumulative_d4(rr, ss))
    # per ray: indices ascending; the class-ordered double sum of the per-hit integrals (class =
    # (index >> 10) & 7, each class in hit order, the 8 class sums added pairwise) IS the ray's
    # column density (same doubles, same order)
    o = np.append(offs.cpu().numpy(), len(idx)); ih = idx.cpu().numpy(); wh = w.cpu().numpy(); dh = dist.cpu().numpy()
    for r in range(0, n_rays, 37):
        seg = slice(o[r], o[r + 1])
        assert np.all(np.diff(ih[seg]) > 0)
        cls = [0.0] * 8
        for i, x in zip(ih[seg], wh[seg]):
            cls[(int(i) >> 10) & 7] += float(x)
        acc = ((cls[0] + cls[1]) + (cls[2] + cls[3])) + ((cls[4] + cls[5]) + (cls[6] + cls[7]))
        assert acc == cu.cpu().numpy()[r]
        assert np.all((dh[seg] >= 0) & (dh[seg] < 2.0))
    # sort_by_distance<double>: == stable per-segment sort
    gh.sort_by_distance(dist, offs, idx, w)
    d2 = dist.cpu().numpy(); i2 = idx.cpu().numpy(); w2 = w.cpu().numpy()
    for r in range(0, n_rays, 11):
        seg = slice(o[r], o[r + 1])
        order = np.argsort(dh[seg], kind="stable")
        assert np.array_equal(d2[seg], dh[seg][order]) and np.array_equal(i2[seg], ih[seg][order])
        assert np.array_equal(w2[seg], wh[seg][order])


# ---- the int offsets of trace_sph -----------------------------------------------------
def test_per_hit_offsets_refuse_totals_beyond_int32(gh, cuda):
    """Synthetic hit counts summing past 2^31: the scan reports the true 64-bit total and the
    wrapper refuses it (the reference's int scan would wrap and the per-hit pass would then
    write below its buffers)."""
    counts = torch.full((1_000_000,), 3000, dtype=torch.int32, device=cuda)
    total = gh.exclusive_scan(counts.clone(), torch.empty_like(counts))
    assert total == 3_000_000_000
    with pytest.raises(ValueError, match="INT32_MAX"):
        gh._offsets_from_counts(counts.clone())
    ok = torch.full((1000,), 2_000_000, dtype=torch.int32, device=cuda)
    assert gh._offsets_from_counts(ok) == 2_000_000_000
    with pytest.raises(ValueError):
        gh._offsets_from_counts(torch.full((1000,), 2_000_000, dtype=torch.int32, device=cuda), extra=2 ** 31)


def test_zero_ray_trace_is_a_no_op(gh, oracle, cuda):
    s = _dev(oracle.random_real4(5000, (0, 0, 0, 0.01), (1, 1, 1, 0.05)), cuda)
    tree = gh.Tree(5000, 32, device=cuda); gh.build_tree(s, tree, (0, 0, 0), (1, 1, 1))
    rays = torch.empty((0, 7), dtype=torch.float32, device=cuda)
    out = torch.empty(0, dtype=torch.float32, device=cuda)
    gh.trace_cumulative_sph(rays, s, tree, out, check=True)
    gh.trace_hitcounts_sph(rays, s, tree, torch.empty(0, dtype=torch.int32, device=cuda))


# ---- a real two-rank sharded trace (both ranks on this one GPU, gathered under gloo) ---------------
def _shard_worker(rank, world, n_side, port, q):
    import torch.distributed as dist
    for p in (os.path.join(ROOT, "grace-devel_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import grace_hip as gh
    import oracle as O
    from grace_hip import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cuda = torch.device("cuda:0")
    n = 100_000
    s = torch.from_numpy(O.random_real4(n, (0, 0, 0, 0.004), (1, 1, 1, 0.03))).to(cuda)
    lo4, hi4 = gh.min_max_vec4(s); lo4[3] = hi4[3] = 0
    tree = gh.Tree(n, 32, device=cuda); gh.build_tree(s, tree, lo4[:3], hi4[:3])   # replicated build
    rays, _ = gh.orthogonal_rays_z(n_side, lo4, hi4, device=cuda)
    n_rays = len(rays)
    per = sharding.shard_size(n_rays, world)
    lo, hi = sharding.shard_bounds(n_rays, world, rank)
    mine = torch.zeros(per, dtype=torch.float32, device=cuda)
    gh.trace_cumulative_sph(rays[lo:hi].contiguous(), s, tree, mine[: hi - lo], check=True)   # hi == lo: no-op
    full = sharding.gather_results(mine, n_rays, world, dist)
    whole = torch.empty(n_rays, dtype=torch.float32, device=cuda)
    gh.trace_cumulative_sph(rays, s, tree, whole, check=True)
    ok = bool(torch.equal(full.view(torch.int32), whole.view(torch.int32))) and float(whole.sum()) > 0
    dist.barrier(); dist.destroy_process_group()
    q.put((rank, ok, hi - lo))


@pytest.mark.parametrize("world,n_side", [(2, 64), (3, 8)])
def test_two_rank_sharded_trace_equals_single_rank_image(world, n_side):
    """Every rank really traces its shard (the sharded flow was only tested with a
    stand-in); n_side 8 with 3 ranks leaves the last rank an EMPTY shard (64 rays, shards of 64)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + n_side) % 2000
    procs = [ctx.Process(target=_shard_worker, args=(r, world, n_side, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    assert all(ok for _, ok, _ in res), res
    if n_side == 8:
        assert res[-1][2] == 0 and res[0][2] == 64      # the trailing ranks' shards are empty


def test_calls_on_alternating_streams_share_the_workspace_safely(gh, oracle, cuda):
    """the library's workspace is one bump-allocated buffer; a frame opened on another
    stream than the previous one must wait (device side) for the previous frame's kernels.
    Back-to-back traces of two different ray sets on two torch streams, no host sync between."""
    n = 300_000
    s = _dev(oracle.random_real4(n, (0, 0, 0, 0.002), (1, 1, 1, 0.02)), cuda)
    tree = gh.Tree(n, 32, device=cuda)
    gh.build_tree(s, tree, (0, 0, 0), (1, 1, 1))
    ra = gh.uniform_random_rays(32 * 600, (0.5, 0.5, 0.5), 2.0, seed=5, device=cuda)
    rb = gh.uniform_random_rays(32 * 600, (0.3, 0.6, 0.4), 2.0, seed=6, device=cuda)
    ref_a = torch.empty(len(ra), dtype=torch.float32, device=cuda); gh.trace_cumulative_sph(ra, s, tree, ref_a)
    ref_b = torch.empty(len(rb), dtype=torch.float32, device=cuda); gh.trace_cumulative_sph(rb, s, tree, ref_b)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rep in range(6):
        st, rays = ((s1, ra), (s2, rb))[rep % 2]
        out = torch.empty(len(rays), dtype=torch.float32, device=cuda)
        with torch.cuda.stream(st):
            gh.trace_cumulative_sph(rays, s, tree, out)
        outs.append(out)
    torch.cuda.synchronize()
    for rep, out in enumerate(outs):
        ref = (ref_a, ref_b)[rep % 2]
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), rep
    gh.trace_status()
