"""Round-3 GPU tests: the generic build entry (caller-evaluated boxes, thrust::greater), the
generic extrema, weighted scan in double -- all through the C ABI (ctypes), against the oracle /
numpy."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


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
