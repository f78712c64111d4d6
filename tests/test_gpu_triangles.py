"""GPU parity for the alternate-primitive path (BASELINE configs[4], the reference's
tests/profile_trace_triangle): triangle centroid keys, XOR-delta ALBVH with TriangleAABB,
closest-hit trace == brute force over all triangles."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def heightfield_mesh(gx, gy, seed=7, flat=False):
    """2 * gx * gy triangles {v, e1, e2} over [0,1]^2, z displaced; normals towards +z."""
    rng = np.random.default_rng(seed)
    xs = np.linspace(0, 1, gx + 1, dtype=np.float32); ys = np.linspace(0, 1, gy + 1, dtype=np.float32)
    z = np.zeros((gy + 1, gx + 1), np.float32) if flat else \
        (0.1 * np.sin(6 * xs)[None, :] * np.cos(5 * ys)[:, None] + 0.01 * rng.standard_normal((gy + 1, gx + 1))).astype(np.float32)
    X, Y = np.meshgrid(xs, ys)
    V = np.stack([X, Y, z], -1).astype(np.float32)
    v00 = V[:-1, :-1]; v10 = V[:-1, 1:]; v01 = V[1:, :-1]; v11 = V[1:, 1:]
    t1 = np.concatenate([v00, v10 - v00, v01 - v00], -1).reshape(-1, 9)
    t2 = np.concatenate([v11, v01 - v11, v10 - v11], -1).reshape(-1, 9)
    tris = np.concatenate([t1, t2], 0).astype(np.float32)
    return np.ascontiguousarray(tris[rng.permutation(len(tris))])


def _dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def _cameras(bot, top, fovy_deg, rx, ry):
    # tests/profile_trace_triangle/tris_trace.cu:7-41 (setup_cameras)
    size = top - bot; center = (bot + top) / 2
    fovy = fovy_deg * 3.141 / 180.
    fovx = 2. * np.arctan2(np.tan(fovy / 2.), rx / ry)
    cam_z = center[2] + max(1.1 * size[0] / fovx, 1.1 * size[1] / fovy)
    cams = [(bot[0] - 0.1 * size[0], top[1] + 0.3 * size[1], cam_z),
            (top[0] + 0.1 * size[0], bot[1] - 0.3 * size[1], cam_z),
            (center[0], center[1], cam_z)]
    return cams, center, (0., 1., 0.), fovy, 100. * size[2]


@pytest.mark.parametrize("gx,gy,mpl", [(16, 16, 1), (64, 64, 8), (128, 96, 32)])
def test_triangle_build_identical(gh, oracle, cuda, gx, gy, mpl):
    tris = heightfield_mesh(gx, gy)
    d = _dev(tris, cuda)
    tree = gh.Tree(len(tris), mpl, device=cuda)
    bot, top = gh.build_tree_tris(d, tree)
    rb, rt = oracle.tri_centroid_bounds(tris)
    assert np.array_equal(bot, rb) and np.array_equal(top, rt)
    keys = oracle.morton_keys30_tri(tris, rb, rt)
    keys, st, _ = oracle.sort_by_key(keys, tris)
    st = np.ascontiguousarray(st)
    assert np.array_equal(d.cpu().numpy(), st)
    nodes, leaves, root, _ = oracle.albvh(st, oracle.deltas_xor(keys), mpl, prim_kind=1)
    assert np.array_equal(tree.leaves.cpu().numpy(), leaves)
    assert np.array_equal(tree.nodes.cpu().numpy(), nodes)
    assert int(tree.root_index.item()) == root


def test_flat_triangles_get_inflated_boxes(gh, oracle, cuda):
    """Triangles of zero extent along z take the AABB_EPSILON branch (triangle.cu:21-35)."""
    tris = heightfield_mesh(48, 48)
    flat = tris[:, 0] < 0.5                      # flatten the left half to z = 0.25
    tris[flat, 2] = 0.25; tris[flat, 5] = 0.0; tris[flat, 8] = 0.0
    d = _dev(tris, cuda)
    tree = gh.Tree(len(tris), 4, device=cuda)
    bot, top = gh.build_tree_tris(d, tree)
    keys = oracle.morton_keys30_tri(tris, bot, top)
    keys, st, _ = oracle.sort_by_key(keys, tris)
    st = np.ascontiguousarray(st)
    nodes, leaves, root, _ = oracle.albvh(st, oracle.deltas_xor(keys), 4, prim_kind=1)
    assert np.array_equal(tree.nodes.cpu().numpy(), nodes)
    f = nodes.view(np.float32)
    assert np.all(f[:, 13] > f[:, 12]) and np.all(f[:, 15] > f[:, 14])   # no zero-thickness box


@pytest.mark.parametrize("ray_order", [True, False])
def test_closest_triangle_equals_brute_force(gh, oracle, cuda, ray_order):
    gh.set_ray_reorder(ray_order)
    try:
        tris = heightfield_mesh(96, 64)
        d = _dev(tris, cuda)
        tree = gh.Tree(len(tris), 32, device=cuda)
        bot, top = gh.build_tree_tris(d, tree)
        st = d.cpu().numpy()
        cams, look_at, up, fovy, length = _cameras(bot.astype(np.float64), top.astype(np.float64), 50., 64, 64)
        for cam in cams:
            rays = gh.pinhole_camera_rays(64, 64, cam, look_at, up, fovy, length, device=cuda)
            ref_rays = oracle.pinhole_rays(64, 64, cam, look_at, up, fovy, length)
            got_rays = rays.cpu().numpy()
            assert np.allclose(got_rays, np.frombuffer(ref_rays.tobytes(), np.float32).reshape(-1, 7),
                               rtol=0, atol=2e-7 * max(1.0, length))
            out = torch.empty(len(rays), dtype=torch.int32, device=cuda)
            gh.trace_closest_tri(rays, d, tree, out)
            gh.trace_status()
            ref, _ = oracle.brute_closest_tri(got_rays, st)
            got = out.cpu().numpy()
            assert np.array_equal(got, ref)
            assert (ref >= 0).sum() > len(ref) // 4        # the mesh is actually seen
    finally:
        gh.set_ray_reorder(True)


def test_back_faces_are_culled(gh, oracle, cuda):
    tris = heightfield_mesh(32, 32)
    d = _dev(tris, cuda)
    tree = gh.Tree(len(tris), 8, device=cuda)
    gh.build_tree_tris(d, tree)
    rays = gh.pinhole_camera_rays(32, 32, (0.5, 0.5, -3.0), (0.5, 0.5, 0.0), (0, 1, 0), 0.5, 100., device=cuda)
    out = torch.empty(len(rays), dtype=torch.int32, device=cuda)
    gh.trace_closest_tri(rays, d, tree, out)        # seen from below: every face is a back face
    assert int((out >= 0).sum()) == 0
    ref, _ = oracle.brute_closest_tri(rays.cpu().numpy(), d.cpu().numpy())
    assert np.array_equal(out.cpu().numpy(), ref)


def test_config5_full_size_properties(gh, oracle, cuda):
    """1 048 576-triangle mesh, 1024^2 primary rays (profile_trace_triangle at BASELINE size):
    packet-order independence and brute-force exactness on a 256-ray subset."""
    tris = heightfield_mesh(1024, 512)
    assert len(tris) == 1048576
    d = _dev(tris, cuda)
    tree = gh.Tree(len(tris), 32, device=cuda)
    bot, top = gh.build_tree_tris(d, tree)
    cams, look_at, up, fovy, length = _cameras(bot.astype(np.float64), top.astype(np.float64), 50., 1024, 1024)
    rays = gh.pinhole_camera_rays(1024, 1024, cams[2], look_at, up, fovy, length, device=cuda)
    a = torch.empty(len(rays), dtype=torch.int32, device=cuda)
    b = torch.empty_like(a)
    gh.trace_closest_tri(rays, d, tree, a)
    gh.set_ray_reorder(False)
    gh.trace_closest_tri(rays, d, tree, b)
    gh.set_ray_reorder(True)
    gh.trace_status()
    assert torch.equal(a, b)
    sub = np.linspace(0, len(rays) - 1, 256).astype(np.int64)
    ref, _ = oracle.brute_closest_tri(rays.cpu().numpy()[sub], d.cpu().numpy())
    assert np.array_equal(a.cpu().numpy()[sub], ref)
    assert int((a >= 0).sum()) > len(rays) // 4
