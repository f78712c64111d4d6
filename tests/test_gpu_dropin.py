"""The drop-in boundary proper: callers written against the REFERENCE's include paths
(grace/cuda/build_sph.cuh, trace_sph.cuh, nodes.h, scan.cuh, sort.cuh, gen_rays.cuh,
util/extrema.cuh, device/intersect.cuh, ray.h) and thrust::device_vector types, compiled by
hipcc against include/ and linked with libgrace_hip.so, must give the same bits as the ctypes
path on the same inputs (tests/cpp/dropin_*.hip are authored here: the call sequences of
tests/project_gadget/project_gadget.cu:58-96 and tests/tree_traversal/tree_traversal.cu:40-100)."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "grace-devel_amd", "lib")
pytestmark = pytest.mark.gpu


def build_dropin(tmp_path, name, flags=("-ffp-contract=off",), suffix=""):
    exe = str(tmp_path / (name + suffix))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", *flags,
                           "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "tests", "cpp"),
                           os.path.join(ROOT, "tests", "cpp", name + ".hip"), "-o", exe,
                           "-L" + LIBDIR, "-lgrace_hip", "-Wl,-rpath," + LIBDIR])
    return exe


def test_project_gadget_through_reference_headers(tmp_path, gh, cuda):
    import torch
    from grace_hip import gadget
    n, side = 300_000, 256
    rng = np.random.default_rng(11)
    pos = rng.random((n, 3), dtype=np.float32)
    h = np.full(n, (3 * 48 / (4 * math.pi * n)) ** (1 / 3), np.float32) * (0.8 + 0.4 * rng.random(n, dtype=np.float32))
    snap = str(tmp_path / "snap")
    gadget.write_gadget(snap, pos, h)
    exe = build_dropin(tmp_path, "dropin_project_gadget")
    out = str(tmp_path / "img.f32")
    r = subprocess.run([exe, str(side * side // 32), "32", snap, out, str(tmp_path / "d.bmp")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    img_cpp = np.fromfile(out, np.float32)
    assert len(img_cpp) == side * side and os.path.getsize(str(tmp_path / "d.bmp")) > 54

    s = torch.from_numpy(gadget.read_gadget(snap)).to(cuda)
    lo, hi = gh.min_max_vec4(s)
    lo[3] = hi[3] = 0.0
    tree = gh.Tree(n, 32, device=cuda)
    gh.build_tree(s, tree, lo[:3], hi[:3])
    rays, _ = gh.orthogonal_rays_z(side, lo, hi, device=cuda)
    img = torch.empty(len(rays), dtype=torch.float32, device=cuda)
    gh.trace_cumulative_sph(rays, s, tree, img)
    assert np.array_equal(img.cpu().numpy().view(np.uint32), img_cpp.view(np.uint32)), \
        "thrust-header path and ctypes path differ"
    assert img_cpp.mean() > 0


def test_tree_traversal_through_reference_headers(tmp_path, gh, cuda):
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    n, n_rays = 200_000, 32 * 100
    exe = build_dropin(tmp_path, "dropin_tree_traversal")
    out = str(tmp_path / "counts.i32")
    r = subprocess.run([exe, str(n), str(n_rays // 32), "32", out], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "PASSED" in r.stdout, r.stdout + r.stderr
    counts_cpp = np.fromfile(out, np.int32)
    low, high = (-1e4, -1e4, -1e4, 80.0), (1e4, 1e4, 1e4, 400.0)
    s = torch.from_numpy(O.random_real4(n, low, high)).to(cuda)
    tree = gh.Tree(n, 32, device=cuda)
    gh.build_tree(s, tree, low[:3], high[:3])
    rays = gh.uniform_random_rays(n_rays, (0.0, 0.0, 0.0), 2e4, seed=1234, device=cuda)
    counts = torch.empty(n_rays, dtype=torch.int32, device=cuda)
    gh.trace_hitcounts_sph(rays, s, tree, counts)
    assert np.array_equal(counts.cpu().numpy(), counts_cpp)
    assert counts_cpp.sum() > 0


def test_remaining_instantiations_through_reference_headers(tmp_path):
    """63-bit keys + 64-bit XOR deltas, surface-area deltas, double4 spheres end to end
    (hit counts, column densities, per-hit outputs, sentinels, sort_by_distance, segmented
    scan in double): each compared inside the program with a host brute-force loop."""
    exe = build_dropin(tmp_path, "dropin_types")
    r = subprocess.run([exe, "60000", "40"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("PASSED"), r.stdout + r.stderr


def test_generic_forms_through_reference_headers(tmp_path):
    """grace::morton_keys / compute_deltas / build_ALBVH with stock and caller-defined functors,
    thrust::greater, min_max_x/y/z/w, min/max_vec2/3/4, weighted_exclusive_segmented_scan<double>
    (tests/cpp/dropin_generic.hip): every generic form equals the sphere-specialised one / a host
    loop, bit for bit.  Built twice: with the test suite's -ffp-contract=off and with hipcc's
    default (contraction on) -- the stock functors pin their own arithmetic."""
    for flags, suffix in ((("-ffp-contract=off",), ""), ((), "_default_flags")):
        exe = build_dropin(tmp_path, "dropin_generic", flags, suffix)
        r = subprocess.run([exe, "70000"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.strip().endswith("PASSED"), r.stdout + r.stderr


@pytest.mark.parametrize("variant", ["less", "greater", "thrust_sort"])
def test_caller_defined_triangles_through_generic_forms(tmp_path, gh, cuda, variant):
    """BASELINE config 5's call sequence (tests/profile_trace_triangle/tris_tree.cuh:17-30,
    tris_trace.cu:43-62) with the CALLER's primitive and functors (tests/cpp/dropin_triangles.hip):
    morton_keys(TriCentre) -> sort -> compute_deltas(DeltaXOR) -> build_ALBVH(TriBox) ->
    trace_texref(HitTri, KeepTri, StartRay, RayExit_to_array).  Sorted primitives, tree and closest
    hits equal the library's built-in triangle path (grace_*_tri, the ctypes path) bit for bit."""
    import torch
    from test_gpu_triangles import heightfield_mesh, _cameras
    tris = heightfield_mesh(96, 64)
    tris[:300, 2] = 0.25; tris[:300, 5] = 0.0; tris[:300, 8] = 0.0      # some faces flat in z
    d = torch.from_numpy(np.ascontiguousarray(tris)).to(cuda)
    mpl = 8
    tree = gh.Tree(len(tris), mpl, device=cuda)
    bot, top = gh.build_tree_tris(d, tree)
    cams, look_at, up, fovy, length = _cameras(bot.astype(np.float64), top.astype(np.float64), 50., 64, 64)
    rays = torch.cat([gh.pinhole_camera_rays(64, 64, cam, look_at, up, fovy, length, device=cuda) for cam in cams])
    out = torch.empty(len(rays), dtype=torch.int32, device=cuda)
    gh.trace_closest_tri(rays, d, tree, out)
    gh.trace_status()

    tris.tofile(str(tmp_path / "tris.f32"))
    rays.cpu().numpy().tofile(str(tmp_path / "rays.f32"))
    flags = ("-ffp-contract=off", "-DUSE_THRUST_SORT") if variant == "thrust_sort" else ("-ffp-contract=off",)
    exe = build_dropin(tmp_path, "dropin_triangles", flags, "_" + variant)
    prefix = str(tmp_path / "out")
    r = subprocess.run([exe, str(tmp_path / "tris.f32"), str(tmp_path / "rays.f32"), str(mpl), prefix]
                       + (["greater"] if variant == "greater" else []), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert np.array_equal(np.fromfile(prefix + ".bounds", np.float32), np.concatenate([bot, top]))
    assert np.array_equal(np.fromfile(prefix + ".tris", np.float32).reshape(-1, 9), d.cpu().numpy())
    assert np.array_equal(np.fromfile(prefix + ".leaves", np.int32).reshape(-1, 4), tree.leaves.cpu().numpy())
    assert np.array_equal(np.fromfile(prefix + ".nodes", np.int32).reshape(-1, 16), tree.nodes.cpu().numpy())
    assert int(np.fromfile(prefix + ".root", np.int32)[0]) == int(tree.root_index.item())
    got = np.fromfile(prefix + ".closest", np.int32)
    assert np.array_equal(got, out.cpu().numpy())
    assert (got >= 0).sum() > len(got) // 4
