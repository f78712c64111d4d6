"""Python host-side mirror of the grace:: SPH API over libgrace_hip.so (C ABI, ctypes).

Names, argument meaning and error behaviour follow the reference's user-facing functions
(include/grace/cuda/build_sph.cuh, trace_sph.cuh, scan.cuh, tests/helper/tree.cuh,
tests/helper/rays.cuh); torch tensors play the role of thrust::device_vector (device
memory + streams only -- every computation is a hand-written HIP kernel behind the C ABI).
There is NO CPU fallback: importing this module without the built library raises.
"""
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libgrace_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libgrace_hip.so is not built (%s). Run __graft_entry__.build() or "
        "`make -C grace-devel_amd`. There is no CPU fallback." % LIB_PATH)

_lib = C.CDLL(LIB_PATH)
_lib.grace_last_error.restype = C.c_char_p
_lib.grace_version.restype = C.c_int

GRACE_OK = 0
GRACE_INVALID_ARGUMENT = 1
GRACE_HIP_ERROR = 2
GRACE_OUT_OF_MEMORY = 3
GRACE_STACK_OVERFLOW = 4

RAY_FLOATS = 7  # include/grace/ray.h:5-10
N_TABLE = 51    # include/grace/cuda/trace_sph.cuh:22


class GraceError(RuntimeError):
    """A GPU API failure; the reference prints the error and exit()s (error.h:40-56)."""


def _check(status):
    if status == GRACE_OK:
        return
    msg = _lib.grace_last_error().decode()
    if status == GRACE_INVALID_ARGUMENT:
        raise ValueError(msg)  # std::invalid_argument in the reference
    if status == GRACE_OUT_OF_MEMORY:
        raise MemoryError(msg)
    raise GraceError("status %d: %s" % (status, msg))


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    return C.c_void_p(t.data_ptr())


def _spheres(t):
    assert t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] == 4
    return t


def _rays(t):
    assert t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] == RAY_FLOATS
    return t


def version():
    return _lib.grace_version()


def exported_symbols():
    """Every entry point include/grace_hip.h declares (checked by the CPU tests)."""
    return _lib


# ---------------------------------------------------------------------------------------
# Tree container -- include/grace/cuda/nodes.h:14-58
# ---------------------------------------------------------------------------------------
class Tree:
    """nodes: int32 [n_nodes, 16] (4 x int4/float4 per node), leaves: int32 [n_leaves, 4],
    root_index: device int32[1]; allocated for N leaves then shrunk by the build
    (nodes.h:48-52, albvh.cuh:842-845)."""

    def __init__(self, n_leaves, max_per_leaf=1, device="cuda"):
        self.max_per_leaf = int(max_per_leaf)
        self.nodes = torch.zeros((max(int(n_leaves) - 1, 1), 16), dtype=torch.int32, device=device)
        self.leaves = torch.zeros((int(n_leaves), 4), dtype=torch.int32, device=device)
        self.root_index = torch.zeros(1, dtype=torch.int32, device=device)

    @property
    def n_leaves(self):
        return self.leaves.shape[0]

    @property
    def n_nodes(self):
        return self.leaves.shape[0] - 1


# ---------------------------------------------------------------------------------------
# Build -- include/grace/cuda/build_sph.cuh
# ---------------------------------------------------------------------------------------
def centroid_bounds(spheres):
    bot = (C.c_float * 3)(); top = (C.c_float * 3)()
    _check(_lib.grace_centroid_bounds_f4(_ptr(_spheres(spheres)), C.c_size_t(len(spheres)),
                                         bot, top, _stream()))
    return np.array(bot, np.float32), np.array(top, np.float32)


def min_max_vec4(v):
    """grace::min_vec4 / max_vec4 (util/extrema.cuh) in one pass."""
    lo = (C.c_float * 4)(); hi = (C.c_float * 4)()
    _check(_lib.grace_minmax_f4(_ptr(_spheres(v)), C.c_size_t(len(v)), lo, hi, _stream()))
    return np.array(lo, np.float32), np.array(hi, np.float32)


def morton_keys_sph(spheres, keys, bot=None, top=None, double_bounds=False):
    """build_sph.cuh:19-35.  keys.dtype selects 30-bit (int32 storage of uint32) or 63-bit
    (int64 storage of uint64) keys."""
    _spheres(spheres)
    if bot is None:
        bot, top = centroid_bounds(spheres)
    n = C.c_size_t(len(spheres))
    if keys.dtype == torch.int32:
        b = (C.c_float * 3)(*[float(x) for x in bot]); t = (C.c_float * 3)(*[float(x) for x in top])
        _check(_lib.grace_morton_keys30_f4(_ptr(spheres), n, b, t, _ptr(keys), _stream()))
    elif keys.dtype == torch.int64:
        if double_bounds:
            b = (C.c_double * 3)(*[float(x) for x in bot]); t = (C.c_double * 3)(*[float(x) for x in top])
            _check(_lib.grace_morton_keys63_f4_d3(_ptr(spheres), n, b, t, _ptr(keys), _stream()))
        else:
            b = (C.c_float * 3)(*[float(x) for x in bot]); t = (C.c_float * 3)(*[float(x) for x in top])
            _check(_lib.grace_morton_keys63_f4(_ptr(spheres), n, b, t, _ptr(keys), _stream()))
    else:
        raise ValueError("keys must be int32 (30-bit) or int64 (63-bit) storage")
    return keys


def sort_by_key(keys, values=None, begin_bit=0, end_bit=None, want_perm=False):
    """thrust::sort_by_key contract: stable, ascending, keys and values permuted in place."""
    n = len(keys)
    perm = torch.empty(n, dtype=torch.int32, device=keys.device) if want_perm else None
    vb = 0
    if values is not None:
        assert values.is_contiguous() and len(values) == n
        vb = values.element_size() * (values.numel() // max(n, 1))
    if keys.dtype == torch.int32:
        fn, bits = _lib.grace_sort_pairs_u32, 32
    elif keys.dtype == torch.int64:
        fn, bits = _lib.grace_sort_pairs_u64, 64
    else:
        raise ValueError("keys must be int32/int64 storage of unsigned keys")
    _check(fn(_ptr(keys), _ptr(values), C.c_size_t(n), C.c_int(vb), C.c_int(begin_bit),
              C.c_int(bits if end_bit is None else end_bit), _ptr(perm), _stream()))
    return perm


def set_sort_overflow_hint(enabled):
    """grace_sort_set_overflow_hint: remember per context that the last large sort overflowed."""
    _check(_lib.grace_sort_set_overflow_hint(C.c_int(1 if enabled else 0)))


def morton_keys30_sort_sph(spheres, bot=None, top=None):
    """build_sph.cuh:41-58: keys + stable sort of the spheres by key, in place."""
    keys = torch.empty(len(spheres), dtype=torch.int32, device=spheres.device)
    morton_keys_sph(spheres, keys, bot, top)
    sort_by_key(keys, spheres, 0, 30)
    return keys


def morton_keys63_sort_sph(spheres, bot=None, top=None):
    """build_sph.cuh:65-82."""
    keys = torch.empty(len(spheres), dtype=torch.int64, device=spheres.device)
    morton_keys_sph(spheres, keys, bot, top)
    sort_by_key(keys, spheres, 0, 63)
    return keys


def euclidean_deltas_sph(spheres, deltas):
    """build_sph.cuh:87-94; deltas has len(spheres) + 1 entries."""
    assert len(deltas) == len(spheres) + 1 and deltas.dtype == torch.float32
    _check(_lib.grace_deltas_euclid_f4(_ptr(_spheres(spheres)), C.c_size_t(len(spheres)),
                                       _ptr(deltas), _stream()))
    return deltas


def surface_area_deltas_sph(spheres, deltas):
    """build_sph.cuh:98-105."""
    assert len(deltas) == len(spheres) + 1 and deltas.dtype == torch.float32
    _check(_lib.grace_deltas_area_f4(_ptr(_spheres(spheres)), C.c_size_t(len(spheres)),
                                     _ptr(deltas), _stream()))
    return deltas


def XOR_deltas_sph(keys, deltas):
    """build_sph.cuh:109-114."""
    assert len(deltas) == len(keys) + 1 and deltas.dtype == keys.dtype
    fn = _lib.grace_deltas_xor_u32 if keys.dtype == torch.int32 else _lib.grace_deltas_xor_u64
    _check(fn(_ptr(keys), C.c_size_t(len(keys)), _ptr(deltas), _stream()))
    return deltas


def ALBVH_sph(spheres, deltas, tree):
    """build_sph.cuh:118-124 -> build_ALBVH (albvh.cuh:986-1021); shrinks tree.nodes/leaves."""
    n = len(spheres)
    assert tree.leaves.shape[0] >= n and len(deltas) == n + 1
    n_leaves = C.c_size_t(0)
    fn = {torch.float32: _lib.grace_albvh_build_f4, torch.float64: _lib.grace_albvh_build_f4_f64,
          torch.int32: _lib.grace_albvh_build_f4_u32,
          torch.int64: _lib.grace_albvh_build_f4_u64}.get(deltas.dtype)
    if fn is None:
        raise ValueError("deltas must be float32/float64 or 32/64-bit XOR deltas")
    _check(fn(_ptr(_spheres(spheres)), C.c_size_t(n), _ptr(deltas), C.c_int(tree.max_per_leaf),
              _ptr(tree.nodes), _ptr(tree.leaves), _ptr(tree.root_index), C.byref(n_leaves),
              _stream()))
    tree.leaves = tree.leaves[: n_leaves.value]
    tree.nodes = tree.nodes[: n_leaves.value - 1]
    return tree


PRIM_SPHERE_F4, PRIM_TRIANGLE, PRIM_SPHERE_D4, PRIM_BOX = 0, 1, 2, 3
COMP_LESS, COMP_GREATER = 0, 1
_DELTA_CODES = {torch.float32: 0, torch.float64: 1, torch.int32: 2, torch.int64: 3}


def build_ALBVH(tree, prims, deltas, prim_kind=PRIM_SPHERE_F4, delta_comp=COMP_LESS):
    """The generic grace::build_ALBVH forms (kernels/albvh.cuh:986-1072) through
    grace_albvh_build_ex: prims are float4 / double4 spheres, triangles [n, 9], or -- PRIM_BOX --
    the caller's AABB functor already evaluated, [n, 6] floats {bot xyz, top xyz}; DeltaComp is
    thrust::less (default) or thrust::greater."""
    n = len(prims)
    assert tree.leaves.shape[0] >= n and len(deltas) == n + 1
    if deltas.dtype not in _DELTA_CODES:
        raise ValueError("deltas must be float32/float64 or 32/64-bit XOR deltas")
    n_leaves = C.c_size_t(0)
    _check(_lib.grace_albvh_build_ex(C.c_int(prim_kind), _ptr(prims), C.c_size_t(n),
                                     C.c_int(_DELTA_CODES[deltas.dtype]), _ptr(deltas), C.c_int(delta_comp),
                                     C.c_int(tree.max_per_leaf), _ptr(tree.nodes), _ptr(tree.leaves),
                                     _ptr(tree.root_index), C.byref(n_leaves), _stream()))
    tree.leaves = tree.leaves[: n_leaves.value]
    tree.nodes = tree.nodes[: n_leaves.value - 1]
    return tree


def min_max_components(data, n_comp, first=0):
    """grace::min_max_x/y/z/w, min/max_vec2/3/4 (util/extrema.cuh:190-772): minima and maxima of
    components first .. first + n_comp - 1 of the rows of a 2-D float32 / float64 / int32 tensor."""
    code = {torch.float32: 0, torch.float64: 1, torch.int32: 2}[data.dtype]
    assert data.dim() == 2 and data.is_contiguous() and first + n_comp <= data.shape[1]
    npdt = {0: np.float32, 1: np.float64, 2: np.int32}[code]
    lo = np.zeros(n_comp, npdt); hi = np.zeros(n_comp, npdt)
    elem = data.element_size()
    _check(_lib.grace_minmax_components(C.c_void_p(data.data_ptr() + first * elem), C.c_size_t(len(data)),
                                        C.c_int(code), C.c_int(n_comp), C.c_size_t(data.shape[1] * elem),
                                        lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p), _stream()))
    return lo, hi


def build_tree(spheres, tree, low=None, high=None):
    """tests/helper/tree.cuh:15-43: 30-bit keys, Euclidean deltas, sorts spheres in place."""
    deltas = torch.empty(len(spheres) + 1, dtype=torch.float32, device=spheres.device)
    morton_keys30_sort_sph(spheres, low, high)
    euclidean_deltas_sph(spheres, deltas)
    ALBVH_sph(spheres, deltas, tree)
    return tree


# ---------------------------------------------------------------------------------------
# Trace -- include/grace/cuda/trace_sph.cuh
# ---------------------------------------------------------------------------------------
def _check_rays(rays):
    _rays(rays)
    if len(rays) % 32 != 0:
        # include/grace/cuda/kernels/bintree_trace.cuh:231-238
        raise ValueError("Number of rays must be a multiple of the warp size (32).")


def _trace_args(rays, spheres, tree):
    return (_ptr(rays), C.c_size_t(len(rays)), _ptr(_spheres(spheres)), C.c_size_t(len(spheres)),
            _ptr(tree.nodes), C.c_size_t(tree.n_nodes), _ptr(tree.leaves), _ptr(tree.root_index))


def hit_integrals(b2, h):
    """functors/trace.cuh:181-186 on arrays (device tensors), the traversal's own arithmetic."""
    out = torch.empty_like(b2)
    _check(_lib.grace_hit_integrals_f32(_ptr(b2), _ptr(h), C.c_size_t(len(b2)), _ptr(out), _stream()))
    return out


def enable_kernel_timing(enabled=True):
    _check(_lib.grace_trace_enable_timing(C.c_int(1 if enabled else 0)))


def last_kernel_ms():
    """Duration of the last traversal kernel alone (HIP events on its stream)."""
    ms = C.c_float(0)
    _check(_lib.grace_trace_last_kernel_ms(C.byref(ms)))
    return ms.value


def last_lattice():
    """1 if the last trace ran the origin-lattice instantiation (measurement hook)."""
    v = C.c_int(0)
    _check(_lib.grace_trace_last_lattice(C.byref(v)))
    return v.value


def set_packet_split(k):
    _check(_lib.grace_trace_set_packet_split(C.c_int(int(k))))


def set_packet_width(w):
    _check(_lib.grace_trace_set_packet_width(C.c_int(int(w))))


def set_exact_integrals(enabled):
    """Column-density trace: True = the reference's per-hit arithmetic bit for bit (slower);
    False (default) = hardware sqrt + fp32 table lerp (within the stated 1e-5 tolerance)."""
    _check(_lib.grace_trace_set_exact_integrals(C.c_int(1 if enabled else 0)))


def set_cache_validation(enabled):
    """1 (default): cached scene / ray records are validated by signature before every use."""
    _check(_lib.grace_trace_set_cache_validation(C.c_int(1 if enabled else 0)))


def set_cache_auto(enabled):
    """1 (default): the records of a scene / ray batch given twice in a row are cached."""
    _check(_lib.grace_trace_set_cache_auto(C.c_int(1 if enabled else 0)))


def set_lattice_split(k):
    _check(_lib.grace_trace_set_lattice_split(C.c_int(k)))


def set_hits_staging(enabled):
    _check(_lib.grace_trace_set_hits_staging(C.c_int(1 if enabled else 0)))


class Context:
    """A library context (include/grace_hip.h, "contexts"): workspace, cached scene / ray order,
    knobs and status word of its own, on the device that is current at creation.  Use as
    `with Context(): ...` in the thread that should run on it."""

    def __init__(self):
        self._h = C.c_void_p(0)
        _check(_lib.grace_context_create(C.byref(self._h)))

    def make_current(self):
        _check(_lib.grace_context_set_current(self._h))

    @staticmethod
    def reset_current():
        _check(_lib.grace_context_set_current(C.c_void_p(0)))

    def destroy(self):
        if self._h:
            _check(_lib.grace_context_destroy(self._h))
            self._h = C.c_void_p(0)

    def __enter__(self):
        self.make_current()
        return self

    def __exit__(self, *exc):
        Context.reset_current()
        self.destroy()
        return False


def set_treelet_size(n):
    _check(_lib.grace_trace_set_treelet_size(C.c_int(int(n))))


def set_ray_reorder(enabled):
    _check(_lib.grace_trace_set_ray_reorder(C.c_int(1 if enabled else 0)))


def trace_status():
    _check(_lib.grace_trace_status(_stream()))


_prepared = {}   # tensors behind the prepared scene / ray batch (kept alive: see trace_prepare)


def trace_prepare(spheres, tree):
    """Computes the scene-constant trace data (per-sphere records, node spans, cluster boxes)
    once; later trace calls over the same spheres / tree reuse it until trace_release().  The
    caller must not modify spheres or tree in between (this library's own sort / build calls
    drop it themselves).  Not part of the reference API (its traces are stateless)."""
    _check(_lib.grace_trace_prepare_f4(_ptr(_spheres(spheres)), C.c_size_t(len(spheres)),
                                       _ptr(tree.nodes), C.c_size_t(tree.n_nodes),
                                       _ptr(tree.leaves), _stream()))
    # The cache is keyed on device pointers: keep the tensors alive while it is, so that the
    # allocator cannot hand their addresses to other data.
    _prepared["scene"] = (spheres, tree.nodes, tree.leaves)


def trace_prepare_tri(tris, tree):
    _check(_lib.grace_trace_prepare_tri(_ptr(_tris(tris)), C.c_size_t(len(tris)),
                                        _ptr(tree.nodes), C.c_size_t(tree.n_nodes),
                                        _ptr(tree.leaves), _stream()))
    _prepared["scene"] = (tris, tree.nodes, tree.leaves)


def trace_prepare_rays(rays):
    """Compute the coherence order of this ray batch once; later traces of the same tensor reuse it."""
    _check(_lib.grace_trace_prepare_rays(_ptr(_rays(rays)), C.c_size_t(len(rays)), _stream()))
    _prepared["rays"] = rays       # (see trace_prepare)


def trace_release_rays():
    _check(_lib.grace_trace_release_rays())
    _prepared.pop("rays", None)


def trace_release():
    _check(_lib.grace_trace_release())
    _prepared.pop("scene", None)


def _trace_hitcounts_keep(rays, spheres, tree, hit_counts):
    """The hit-count pass of trace_sph / trace_with_sentinels_sph: for small batches the library
    keeps the hits per (ray, primitive chunk) for the per-hit pass that follows."""
    _check_rays(rays)
    _check(_lib.grace_trace_hitcounts_keep_f4(*_trace_args(rays, spheres, tree), _ptr(hit_counts),
                                              _stream()))
    return hit_counts


def trace_hitcounts_sph(rays, spheres, tree, hit_counts, check=False):
    """trace_sph.cuh:58-80.  check=True also reads the traversal's status word (a
    synchronisation): packet-stack exhaustion then raises instead of waiting for the next
    trace_status() call."""
    _check_rays(rays)
    assert hit_counts.dtype == torch.int32 and len(hit_counts) == len(rays)
    _check(_lib.grace_trace_hitcounts_f4(*_trace_args(rays, spheres, tree), _ptr(hit_counts),
                                         _stream()))
    if check:
        trace_status()
    return hit_counts


def trace_cumulative_sph(rays, spheres, tree, cumulated, check=False):
    """trace_sph.cuh:82-110.  Asynchronous by default (the bench's timed loop relies on it);
    check=True reads the status word after the launch, as the C++ mirrors do."""
    _check_rays(rays)
    assert cumulated.dtype == torch.float32 and len(cumulated) == len(rays)
    _check(_lib.grace_trace_cumulative_f4(*_trace_args(rays, spheres, tree), _ptr(cumulated),
                                          _stream()))
    if check:
        trace_status()
    return cumulated


INT32_MAX = 2 ** 31 - 1


def _offsets_from_counts(offsets, extra=0):
    """Hit counts -> exclusive offsets in place; returns the total (64-bit).  int offsets cannot
    address more than INT32_MAX per-hit slots: ValueError (std::invalid_argument in the C++
    mirrors) instead of the reference's silent wrap-around."""
    total = exclusive_scan(offsets, offsets)
    if total + extra > INT32_MAX:
        raise ValueError("trace_sph: %d hits (+ %d sentinels) exceed INT32_MAX; the int ray offsets "
                         "cannot address the per-hit arrays. Trace fewer rays per call." % (total, extra))
    return total


def trace_sph(rays, spheres, tree):
    """trace_sph.cuh:112-168: returns (ray_offsets, hit_indices, hit_integrals,
    hit_distances); the reference resizes the three per-hit vectors to the total."""
    _check_rays(rays)
    n = len(rays)
    offsets = torch.empty(n, dtype=torch.int32, device=rays.device)
    _trace_hitcounts_keep(rays, spheres, tree, offsets)
    total = _offsets_from_counts(offsets)
    idx = torch.empty(total, dtype=torch.int32, device=rays.device)
    integrals = torch.empty(total, dtype=torch.float32, device=rays.device)
    dists = torch.empty(total, dtype=torch.float32, device=rays.device)
    if total == 0:             # no ray hits anything: empty outputs, like the reference's resize(0)
        return offsets, idx, integrals, dists
    _check(_lib.grace_trace_hits_f4(*_trace_args(rays, spheres, tree), _ptr(offsets), _ptr(idx),
                                    _ptr(integrals), _ptr(dists), _stream()))
    trace_status()
    return offsets, idx, integrals, dists


def trace_with_sentinels_sph(rays, spheres, tree, index_sentinel, integral_sentinel,
                             distance_sentinel):
    """trace_sph.cuh:171-241: like trace_sph, but every ray's segment ends with one sentinel
    slot; returns (ray_offsets, hit_indices, hit_integrals, hit_distances)."""
    _check_rays(rays)
    n = len(rays)
    offsets = torch.empty(n, dtype=torch.int32, device=rays.device)
    _trace_hitcounts_keep(rays, spheres, tree, offsets)
    total = _offsets_from_counts(offsets, extra=n) + n
    _check(_lib.grace_add_iota_i32(_ptr(offsets), C.c_size_t(n), _stream()))
    idx = torch.empty(total, dtype=torch.int32, device=rays.device)
    integrals = torch.empty(total, dtype=torch.float32, device=rays.device)
    dists = torch.empty(total, dtype=torch.float32, device=rays.device)
    bits = lambda f: int(np.float32(f).view(np.uint32))
    _check(_lib.grace_fill_u32(_ptr(idx), C.c_size_t(total), C.c_uint32(index_sentinel & 0xFFFFFFFF), _stream()))
    _check(_lib.grace_fill_u32(_ptr(integrals), C.c_size_t(total), C.c_uint32(bits(integral_sentinel)), _stream()))
    _check(_lib.grace_fill_u32(_ptr(dists), C.c_size_t(total), C.c_uint32(bits(distance_sentinel)), _stream()))
    _check(_lib.grace_trace_hits_f4(*_trace_args(rays, spheres, tree), _ptr(offsets), _ptr(idx),
                                    _ptr(integrals), _ptr(dists), _stream()))
    trace_status()
    return offsets, idx, integrals, dists


def trace_stats(rays, spheres, tree):
    """Per ray {nodes, leaves, spheres tested, hits} for that ray alone."""
    stats = torch.empty((len(rays), 4), dtype=torch.int32, device=rays.device)
    _check(_lib.grace_trace_stats_f4(*_trace_args(rays, spheres, tree), _ptr(stats), _stream()))
    return stats


# ---------------------------------------------------------------------------------------
# Triangles -- tests/profile_trace_triangle/{tris_tree.cuh,tris_trace.cu}
# ---------------------------------------------------------------------------------------
def _tris(t):
    assert t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] == 9 and t.is_contiguous()
    return t


def build_tree_tris(tris, tree):
    """tris_tree.cuh:17-30: centroid bounds, 30-bit keys, stable sort of the triangles, XOR
    deltas, ALBVH with TriangleAABB.  Sorts tris in place; returns (bots, tops)."""
    n = len(_tris(tris))
    bot = (C.c_float * 3)(); top = (C.c_float * 3)()
    _check(_lib.grace_centroid_bounds_tri(_ptr(tris), C.c_size_t(n), bot, top, _stream()))
    keys = torch.empty(n, dtype=torch.int32, device=tris.device)
    _check(_lib.grace_morton_keys30_tri(_ptr(tris), C.c_size_t(n), bot, top, _ptr(keys), _stream()))
    sort_by_key(keys, tris, 0, 30)
    deltas = torch.empty(n + 1, dtype=torch.int32, device=tris.device)
    XOR_deltas_sph(keys, deltas)
    n_leaves = C.c_size_t(0)
    _check(_lib.grace_albvh_build_tri_u32(_ptr(tris), C.c_size_t(n), _ptr(deltas),
                                          C.c_int(tree.max_per_leaf), _ptr(tree.nodes),
                                          _ptr(tree.leaves), _ptr(tree.root_index),
                                          C.byref(n_leaves), _stream()))
    tree.leaves = tree.leaves[: n_leaves.value]
    tree.nodes = tree.nodes[: n_leaves.value - 1]
    return np.array(bot, np.float32), np.array(top, np.float32)


def trace_closest_tri(rays, tris, tree, closest):
    """tris_trace.cu:43-62."""
    _check_rays(rays)
    assert closest.dtype == torch.int32 and len(closest) == len(rays)
    _check(_lib.grace_trace_closest_tri(_ptr(rays), C.c_size_t(len(rays)), _ptr(_tris(tris)),
                                        C.c_size_t(len(tris)), _ptr(tree.nodes),
                                        C.c_size_t(tree.n_nodes), _ptr(tree.leaves),
                                        _ptr(tree.root_index), _ptr(closest), _stream()))
    trace_status()
    return closest


def pinhole_camera_rays(res_x, res_y, camera, look_at, view_up, fovy, length, device="cuda"):
    rays = torch.empty((res_x * res_y, RAY_FLOATS), dtype=torch.float32, device=device)
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    _check(_lib.grace_rays_pinhole(C.c_int(res_x), C.c_int(res_y), f3(camera), f3(look_at),
                                   f3(view_up), C.c_float(fovy), C.c_float(length), _ptr(rays),
                                   _stream()))
    return rays


# ---------------------------------------------------------------------------------------
# Scans -- include/grace/cuda/scan.cuh
# ---------------------------------------------------------------------------------------
def exclusive_scan(inp, out):
    total = C.c_longlong(0)
    _check(_lib.grace_scan_exclusive_i32(_ptr(inp), C.c_size_t(len(inp)), _ptr(out),
                                         C.byref(total), _stream()))
    return total.value


def exclusive_segmented_scan(segment_offsets, data, results):
    """scan.cuh:15-37; data and results may be the same tensor."""
    assert segment_offsets.dtype == torch.int32 and data.dtype == results.dtype
    fn = {torch.float32: _lib.grace_segscan_exclusive_f32,
          torch.float64: _lib.grace_segscan_exclusive_f64}[data.dtype]
    _check(fn(_ptr(segment_offsets), C.c_size_t(len(segment_offsets)), _ptr(data),
              C.c_size_t(len(data)), _ptr(results), _stream()))
    return results


def sort_by_distance(hit_distances, ray_offsets, hit_indices, hit_data):
    """sort.cuh:100-131: per-ray sort by distance; indices and data follow."""
    fn = _lib.grace_sort_by_distance_f64 if hit_distances.dtype == torch.float64 \
        else _lib.grace_sort_by_distance_f32
    assert hit_data is None or hit_data.dtype == hit_distances.dtype
    _check(fn(_ptr(hit_distances), _ptr(ray_offsets), C.c_size_t(len(ray_offsets)),
              C.c_size_t(len(hit_distances)), _ptr(hit_indices), _ptr(hit_data), _stream()))


def weighted_exclusive_segmented_scan(to_sum, weights, weight_map, segment_offsets, out):
    """scan.cuh:43-58."""
    weighted = torch.empty_like(to_sum)
    fn = _lib.grace_multiply_by_weights_f64 if to_sum.dtype == torch.float64 else _lib.grace_multiply_by_weights_f32
    _check(fn(_ptr(to_sum), C.c_size_t(len(to_sum)),
              _ptr(weights), _ptr(weight_map), _ptr(weighted), _stream()))
    return exclusive_segmented_scan(segment_offsets, weighted, out)


# ---------------------------------------------------------------------------------------
# Rays -- tests/helper/rays.cuh, include/grace/cuda/gen_rays.cuh
# ---------------------------------------------------------------------------------------
def orthogonal_rays_z(n_side, mins4, maxs4, device="cuda"):
    """tests/helper/rays.cuh:55-79; returns (rays [n_side^2, 7], area per ray)."""
    rays = torch.empty((n_side * n_side, RAY_FLOATS), dtype=torch.float32, device=device)
    lo = (C.c_float * 4)(*[float(x) for x in mins4]); hi = (C.c_float * 4)(*[float(x) for x in maxs4])
    area = C.c_float(0)
    _check(_lib.grace_rays_orthogonal_z(C.c_int(n_side), lo, hi, _ptr(rays), C.byref(area),
                                        _stream()))
    return rays, area.value


def healpix_rays(nside, origin, length, device="cuda"):
    rays = torch.empty((12 * nside * nside, RAY_FLOATS), dtype=torch.float32, device=device)
    _check(_lib.grace_rays_healpix(C.c_int(nside), C.c_float(origin[0]), C.c_float(origin[1]),
                                   C.c_float(origin[2]), C.c_float(length), _ptr(rays), _stream()))
    return rays


def uniform_random_rays(n_rays, origin, length, seed=1234, device="cuda"):
    """gen_rays.cuh uniform_random_rays: isotropic directions, direction-Morton sorted."""
    rays = torch.empty((n_rays, RAY_FLOATS), dtype=torch.float32, device=device)
    _check(_lib.grace_rays_isotropic(C.c_size_t(n_rays), C.c_float(origin[0]),
                                     C.c_float(origin[1]), C.c_float(origin[2]),
                                     C.c_float(length), C.c_uint64(seed), _ptr(rays), _stream()))
    return rays


# enum Octants / enum RaySortType, grace/types.h:36-51
PPP, PPM, PMP, PMM, MPP, MPM, MMP, MMM = 7, 6, 5, 4, 3, 2, 1, 0
NoSort, DirectionSort, EndPointSort = 0, 1, 2


def uniform_random_rays_single_octant(n_rays, origin, length, octant=PPP, seed=1234, device="cuda"):
    """gen_rays.cuh:62-97: isotropic directions confined to one octant, direction-sorted."""
    rays = torch.empty((n_rays, RAY_FLOATS), dtype=torch.float32, device=device)
    _check(_lib.grace_rays_isotropic_octant(C.c_size_t(n_rays), C.c_float(origin[0]),
                                            C.c_float(origin[1]), C.c_float(origin[2]),
                                            C.c_float(length), C.c_int(int(octant)),
                                            C.c_uint64(seed), _ptr(rays), _stream()))
    return rays


def _points(points):
    assert points.is_contiguous() and points.dim() == 2 and 3 <= points.shape[1] <= 16
    assert points.dtype in (torch.float32, torch.float64)
    return C.c_int(1 if points.dtype == torch.float64 else 0), C.c_int(points.shape[1])


def one_to_many_rays(origin, points, sort_type=DirectionSort, bot=None, top=None):
    """gen_rays.cuh:99-208: one ray from `origin` to each point ([n, 3..] float32/float64).
    EndPointSort needs the points' bounds (computed here when not given -- the reference's
    bounds-free overload passes AABB_bot twice, which is not reproduced)."""
    isd, k = _points(points)
    rays = torch.empty((len(points), RAY_FLOATS), dtype=torch.float32, device=points.device)
    b = t = None
    if sort_type == EndPointSort:
        if bot is None:
            bot = points[:, :3].min(dim=0).values.float().tolist()
            top = points[:, :3].max(dim=0).values.float().tolist()
        b = (C.c_float * 3)(*[float(x) for x in bot]); t = (C.c_float * 3)(*[float(x) for x in top])
    _check(_lib.grace_rays_one_to_many(C.c_size_t(len(points)), C.c_float(origin[0]),
                                       C.c_float(origin[1]), C.c_float(origin[2]), _ptr(points),
                                       isd, k, C.c_int(int(sort_type)), b, t, _ptr(rays), _stream()))
    return rays


def plane_parallel_random_rays(width, height, base, w, h, length, seed=1234, device="cuda"):
    """gen_rays.cuh:210-262: one ray per cell of the width x height grid spanned by w, h."""
    rays = torch.empty((width * height, RAY_FLOATS), dtype=torch.float32, device=device)
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    _check(_lib.grace_rays_plane_parallel_random(C.c_int(width), C.c_int(height), f3(base), f3(w),
                                                 f3(h), C.c_float(length), C.c_uint64(seed),
                                                 _ptr(rays), _stream()))
    return rays


def orthographic_projection_rays(res_x, res_y, camera, look_at, view_up, vertical_extent, length,
                                 device="cuda"):
    """gen_rays.cuh:264-329."""
    rays = torch.empty((res_x * res_y, RAY_FLOATS), dtype=torch.float32, device=device)
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    _check(_lib.grace_rays_orthographic_projection(C.c_int(res_x), C.c_int(res_y), f3(camera),
                                                   f3(look_at), f3(view_up),
                                                   C.c_float(vertical_extent), C.c_float(length),
                                                   _ptr(rays), _stream()))
    return rays


def morton_keys_points(points, keys, bot, top):
    """morton_keys over float3/float4/double3/double4 points (build_sph.cuh:16-33 with
    Real4 = double4; tests/morton_key_kernel/63bit_keys.cu): co-ordinates narrowed to float
    first.  keys.dtype int32 -> 30-bit, int64 -> 63-bit."""
    isd, k = _points(points)
    b = (C.c_float * 3)(*[float(x) for x in bot]); t = (C.c_float * 3)(*[float(x) for x in top])
    fn = _lib.grace_morton_keys30_points if keys.dtype == torch.int32 else _lib.grace_morton_keys63_points
    _check(fn(_ptr(points), C.c_size_t(len(points)), isd, k, b, t, _ptr(keys), _stream()))
    return keys


# ---------------------------------------------------------------------------------------
# double4 spheres (Real4 = double4, Real = double)
# ---------------------------------------------------------------------------------------
def _spheres_d4(s):
    assert s.is_cuda and s.is_contiguous() and s.dtype == torch.float64 and s.dim() == 2 and s.shape[1] == 4


def build_tree_d4(spheres, tree, bot, top):
    """tests/helper/tree.cuh build_tree with double4 spheres: 30-bit keys of the float-narrowed
    centres, sort of the 32-byte records, Euclidean deltas (formed in double, stored as float),
    ALBVH.  Sorts `spheres` in place."""
    _spheres_d4(spheres)
    n = len(spheres)
    keys = torch.empty(n, dtype=torch.int32, device=spheres.device)
    morton_keys_points(spheres, keys, bot, top)
    sort_by_key(keys, spheres, 0, 30)
    deltas = torch.empty(n + 1, dtype=torch.float32, device=spheres.device)
    _check(_lib.grace_deltas_euclid_d4(_ptr(spheres), C.c_size_t(n), _ptr(deltas), _stream()))
    n_leaves = C.c_size_t(0)
    _check(_lib.grace_albvh_build_d4(_ptr(spheres), C.c_size_t(n), _ptr(deltas),
                                     C.c_int(tree.max_per_leaf), _ptr(tree.nodes), _ptr(tree.leaves),
                                     _ptr(tree.root_index), C.byref(n_leaves), _stream()))
    tree.leaves = tree.leaves[: n_leaves.value]
    tree.nodes = tree.nodes[: n_leaves.value - 1]
    return deltas


def _trace_args_d4(rays, spheres, tree):
    _check_rays(rays); _spheres_d4(spheres)
    return (_ptr(rays), C.c_size_t(len(rays)), _ptr(spheres), C.c_size_t(len(spheres)),
            _ptr(tree.nodes), C.c_size_t(tree.n_leaves - 1), _ptr(tree.leaves), _ptr(tree.root_index))


def trace_hitcounts_d4(rays, spheres, tree, counts):
    _check(_lib.grace_trace_hitcounts_d4(*_trace_args_d4(rays, spheres, tree), _ptr(counts), _stream()))
    _check(_lib.grace_trace_status_d4(_stream()))
    return counts


def trace_cumulative_d4(rays, spheres, tree, sums):
    assert sums.dtype == torch.float64
    _check(_lib.grace_trace_cumulative_d4(*_trace_args_d4(rays, spheres, tree), _ptr(sums), _stream()))
    _check(_lib.grace_trace_status_d4(_stream()))
    return sums


def surface_area_deltas_d4(spheres, deltas):
    """surface_area_deltas_sph<double4> (build_sph.cuh:97-105); deltas float32 or float64."""
    _spheres_d4(spheres)
    fn = _lib.grace_deltas_area_d4_f64 if deltas.dtype == torch.float64 else _lib.grace_deltas_area_d4
    _check(fn(_ptr(spheres), C.c_size_t(len(spheres)), _ptr(deltas), _stream()))
    return deltas


def euclidean_deltas_d4(spheres, deltas):
    _spheres_d4(spheres)
    fn = _lib.grace_deltas_euclid_d4_f64 if deltas.dtype == torch.float64 else _lib.grace_deltas_euclid_d4
    _check(fn(_ptr(spheres), C.c_size_t(len(spheres)), _ptr(deltas), _stream()))
    return deltas


def ALBVH_d4(spheres, deltas, tree):
    """ALBVH_sph<double4, DeltaType> for float / double / 32- / 64-bit XOR deltas."""
    _spheres_d4(spheres)
    fn = {torch.float32: _lib.grace_albvh_build_d4, torch.float64: _lib.grace_albvh_build_d4_f64,
          torch.int32: _lib.grace_albvh_build_d4_u32, torch.int64: _lib.grace_albvh_build_d4_u64}[deltas.dtype]
    n_leaves = C.c_size_t(0)
    _check(fn(_ptr(spheres), C.c_size_t(len(spheres)), _ptr(deltas), C.c_int(tree.max_per_leaf),
              _ptr(tree.nodes), _ptr(tree.leaves), _ptr(tree.root_index), C.byref(n_leaves), _stream()))
    tree.leaves = tree.leaves[: n_leaves.value]
    tree.nodes = tree.nodes[: n_leaves.value - 1]
    return tree


def trace_sph_d4(rays, spheres, tree):
    """trace_sph<double4, int, double> (trace_sph.cuh:112-168): (ray_offsets, hit_indices,
    hit_integrals float64, hit_distances float64)."""
    n = len(rays)
    offsets = torch.empty(n, dtype=torch.int32, device=rays.device)
    trace_hitcounts_d4(rays, spheres, tree, offsets)
    total = _offsets_from_counts(offsets)
    idx = torch.empty(total, dtype=torch.int32, device=rays.device)
    integrals = torch.empty(total, dtype=torch.float64, device=rays.device)
    dists = torch.empty(total, dtype=torch.float64, device=rays.device)
    if total:
        _check(_lib.grace_trace_hits_d4(*_trace_args_d4(rays, spheres, tree), _ptr(offsets), _ptr(idx),
                                        _ptr(integrals), _ptr(dists), _stream()))
        trace_status()
    return offsets, idx, integrals, dists


def project_sph(spheres, n_side, max_per_leaf=32):
    """The projection of tests/project_gadget/project_gadget.cu:58-81: bounds with
    w = 0, build_tree, orthogonal_rays_z, trace_cumulative_sph.  Sorts spheres in place.
    Returns (image [n_side, n_side] float32, tree, rays)."""
    lo, hi = min_max_vec4(spheres)
    lo[3] = 0.0; hi[3] = 0.0
    tree = Tree(len(spheres), max_per_leaf, device=spheres.device)
    build_tree(spheres, tree, lo[:3], hi[:3])
    rays, _ = orthogonal_rays_z(n_side, lo, hi, device=spheres.device)
    out = torch.empty(len(rays), dtype=torch.float32, device=spheres.device)
    trace_cumulative_sph(rays, spheres, tree, out)
    return out.view(n_side, n_side), tree, rays
