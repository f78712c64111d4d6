"""ctypes/numpy front-end of the CPU oracle (oracle/grace_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product path (grace-devel_amd/).
Each wrapper names the oracle C function, whose comment cites the reference file:line.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgrace_oracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libchealpix_ref.so")

RAY_DTYPE = np.dtype([("dx", "f4"), ("dy", "f4"), ("dz", "f4"),
                      ("ox", "f4"), ("oy", "f4"), ("oz", "f4"), ("length", "f4")])
assert RAY_DTYPE.itemsize == 28


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "grace_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.go_albvh_leaves_f32.restype = C.c_long
        _lib.go_albvh_leaves_u32.restype = C.c_long
        _lib.go_albvh_leaves_u64.restype = C.c_long
        _lib.go_exclusive_scan_i32.restype = C.c_long
        _lib.go_space_by_two_10bit.restype = C.c_uint32
        _lib.go_space_by_two_10bit.argtypes = [C.c_uint32]
        _lib.go_space_by_two_21bit.restype = C.c_uint64
        _lib.go_space_by_two_21bit.argtypes = [C.c_uint64]
        _lib.go_morton_key30.restype = C.c_uint32
        _lib.go_morton_key30.argtypes = [C.c_uint32] * 3
        _lib.go_morton_key63.restype = C.c_uint64
        _lib.go_morton_key63.argtypes = [C.c_uint64] * 3
        _lib.go_morton_key30_unit.restype = C.c_uint32
        _lib.go_morton_key30_unit.argtypes = [C.c_float] * 3
        _lib.go_morton_key63_unit.restype = C.c_uint64
        _lib.go_morton_key63_unit.argtypes = [C.c_double] * 3
        _lib.go_hit_integral.restype = C.c_float
        _lib.go_hit_integral.argtypes = [C.c_float, C.c_float]
        _lib.go_hash.restype = C.c_uint32
        _lib.go_hash.argtypes = [C.c_uint32]
    return _lib


def ref_healpix():
    """The reference's own chealpix.c compiled under oracle/_ref, or None."""
    if not os.path.exists(_REF_PATH):
        return None
    r = C.CDLL(_REF_PATH)
    r.pix2vec_nest.argtypes = [C.c_long, C.c_long, C.POINTER(C.c_double)]
    r.pix2vec_nest.restype = None
    r.nside2npix.argtypes = [C.c_long]
    r.nside2npix.restype = C.c_long
    return r


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f4(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 4
    return a


def _rays(r):
    r = np.ascontiguousarray(r)
    if r.dtype != RAY_DTYPE:
        r = np.ascontiguousarray(r, dtype=np.float32)
        assert r.ndim == 2 and r.shape[1] == 7
    return r


# -- Morton -------------------------------------------------------------------------

def morton_keys30(prims, bot, top, all_threads=False, out=None):
    prims = _f4(prims)
    keys = np.empty(len(prims), np.uint32) if out is None else out
    b = np.asarray(bot, np.float32); t = np.asarray(top, np.float32)
    fn = lib().go_morton_keys30_f4_omp if all_threads else lib().go_morton_keys30_f4
    fn(_p(prims), C.c_size_t(len(prims)), _p(b), _p(t), _p(keys))
    return keys


def morton_keys63(prims, bot, top, double_bounds=False):
    prims = _f4(prims)
    keys = np.empty(len(prims), np.uint64)
    if double_bounds:
        b = np.asarray(bot, np.float64); t = np.asarray(top, np.float64)
        lib().go_morton_keys63_f4_d3(_p(prims), C.c_size_t(len(prims)), _p(b), _p(t), _p(keys))
    else:
        b = np.asarray(bot, np.float32); t = np.asarray(top, np.float32)
        lib().go_morton_keys63_f4(_p(prims), C.c_size_t(len(prims)), _p(b), _p(t), _p(keys))
    return keys


def centroid_bounds(prims):
    prims = _f4(prims)
    b = np.empty(3, np.float32); t = np.empty(3, np.float32)
    lib().go_centroid_bounds_f4(_p(prims), C.c_size_t(len(prims)), _p(b), _p(t))
    return b, t


def random_real4(n, lo, hi, first=0):
    out = np.empty((n, 4), np.float32)
    lo = np.asarray(lo, np.float32); hi = np.asarray(hi, np.float32)
    lib().go_random_real4(C.c_uint32(first), C.c_size_t(n), _p(lo), _p(hi), _p(out))
    return out


def sort_by_key(keys, values):
    """thrust::sort_by_key contract (build_sph.cuh:46): stable, ascending."""
    order = np.argsort(keys, kind="stable")
    return keys[order], values[order], order


# -- deltas -------------------------------------------------------------------------

def deltas_euclid(prims):
    prims = _f4(prims)
    d = np.empty(len(prims) + 1, np.float32)
    lib().go_deltas_euclid_f4(_p(prims), C.c_size_t(len(prims)), _p(d))
    return d


def deltas_area(prims):
    prims = _f4(prims)
    d = np.empty(len(prims) + 1, np.float32)
    lib().go_deltas_area_f4(_p(prims), C.c_size_t(len(prims)), _p(d))
    return d


def deltas_xor(keys):
    keys = np.ascontiguousarray(keys)
    d = np.empty(len(keys) + 1, keys.dtype)
    fn = lib().go_deltas_xor_u32 if keys.dtype == np.uint32 else lib().go_deltas_xor_u64
    fn(_p(keys), C.c_size_t(len(keys)), _p(d))
    return d


# -- tree ---------------------------------------------------------------------------

_SUFFIX = {np.dtype(np.float32): "f32", np.dtype(np.uint32): "u32", np.dtype(np.uint64): "u64"}


def albvh(prims, deltas, max_per_leaf, prim_kind=0):
    """Returns (nodes[n_nodes,16] as int32 view, leaves[n_leaves,4], root)."""
    prims = np.ascontiguousarray(prims, np.float64 if prim_kind == 2 else np.float32)
    deltas = np.ascontiguousarray(deltas)
    sfx = _SUFFIX[deltas.dtype]
    n = len(prims)
    leaves = np.zeros((n, 4), np.int32)
    n_leaves = getattr(lib(), "go_albvh_leaves_" + sfx)(
        _p(deltas), C.c_size_t(n), C.c_int(max_per_leaf), _p(leaves))
    if n_leaves < 0:
        raise ValueError("max_per_leaf must be less than the total number of primitives.")
    leaves = np.ascontiguousarray(leaves[:n_leaves])
    ld = np.empty(n_leaves + 1, deltas.dtype)
    getattr(lib(), "go_leaf_deltas_" + sfx)(_p(leaves), C.c_size_t(n_leaves), _p(deltas), _p(ld))
    nodes = np.zeros((n_leaves - 1, 16), np.int32)
    root = C.c_int(-1)
    rc = getattr(lib(), "go_albvh_nodes_" + sfx)(
        _p(leaves), C.c_size_t(n_leaves), _p(prims), C.c_int(prim_kind), _p(ld), _p(nodes),
        C.byref(root))
    if rc != 0:
        raise RuntimeError("oracle node build found no root")
    return nodes, leaves, root.value, ld


# -- intersection / traversal -------------------------------------------------------

def brute_hitcounts(rays, prims):
    rays = _rays(rays); prims = _f4(prims)
    out = np.empty(len(rays), np.int32)
    lib().go_brute_hitcounts(_p(rays), C.c_size_t(len(rays)), _p(prims),
                             C.c_size_t(len(prims)), _p(out))
    return out


SUM_BLOCKS = 8   # the class-ordered pairwise fp32 sum this implementation states (grace_oracle.c)


def brute_cumulative(rays, prims, blocks=SUM_BLOCKS):
    """(fp32 class-ordered pairwise sum, fp64 sum).  blocks=1: the reference's single running sum."""
    rays = _rays(rays); prims = _f4(prims)
    out = np.empty(len(rays), np.float32); out64 = np.empty(len(rays), np.float64)
    lib().go_brute_cumulative(_p(rays), C.c_size_t(len(rays)), _p(prims),
                              C.c_size_t(len(prims)), _p(out), _p(out64), C.c_int(blocks))
    return out, out64


def kernel_table():
    """The 51-entry line-integral table (kernel_integrals.h) as float64."""
    n = C.c_int(0)
    lib().go_kernel_table.restype = C.POINTER(C.c_double)
    ptr = lib().go_kernel_table(C.byref(n))
    return np.ctypeslib.as_array(ptr, shape=(n.value,)).copy()


def brute_hits(rays, prims):
    """Returns offsets (exclusive), idx, integrals, distances -- trace_sph contract."""
    counts = brute_hitcounts(rays, prims)
    offsets = np.empty_like(counts)
    total = lib().go_exclusive_scan_i32(_p(counts), C.c_size_t(len(counts)), _p(offsets))
    idx = np.empty(total, np.int32); integ = np.empty(total, np.float32)
    dist = np.empty(total, np.float32)
    rays = _rays(rays); prims = _f4(prims)
    lib().go_brute_hits(_p(rays), C.c_size_t(len(rays)), _p(prims), C.c_size_t(len(prims)),
                        _p(offsets), _p(idx), _p(integ), _p(dist))
    return offsets, idx, integ, dist


def trace(rays, prims, nodes, leaves, root, width=32, mode=0, stats=False, blocks=SUM_BLOCKS):
    rays = _rays(rays); prims = _f4(prims)
    nodes = np.ascontiguousarray(nodes); leaves = np.ascontiguousarray(leaves, np.int32)
    out = np.empty(len(rays), np.int32 if mode == 0 else np.float32)
    st = np.zeros((len(rays), 4), np.uint64) if stats else None
    rc = lib().go_trace(_p(rays), C.c_size_t(len(rays)), _p(prims), C.c_size_t(len(prims)),
                        _p(nodes), C.c_size_t(len(nodes)), _p(leaves), C.c_int(root),
                        C.c_int(width), C.c_int(mode), _p(out),
                        _p(st) if stats else None, C.c_int(blocks))
    if rc != 0:
        raise RuntimeError("oracle traversal stack overflow / bad width")
    return (out, st) if stats else out


def aabbs_hit(ray, node16):
    ray = _rays(ray); node16 = np.ascontiguousarray(node16)
    return lib().go_aabbs_hit(_p(ray), _p(node16))


# -- scans --------------------------------------------------------------------------

def exclusive_scan_i32(a):
    a = np.ascontiguousarray(a, np.int32)
    out = np.empty_like(a)
    total = lib().go_exclusive_scan_i32(_p(a), C.c_size_t(len(a)), _p(out))
    return out, total


def segscan(offsets, data):
    offsets = np.ascontiguousarray(offsets, np.int32)
    data = np.ascontiguousarray(data)
    out = np.zeros_like(data)
    fn = lib().go_segscan_f32 if data.dtype == np.float32 else lib().go_segscan_f64
    fn(_p(offsets), C.c_size_t(len(offsets)), _p(data), C.c_size_t(len(data)), _p(out))
    return out


# -- rays ---------------------------------------------------------------------------

def healpix_dirs(nside):
    n = 12 * nside * nside
    out = np.empty((n, 3), np.float64)
    f = lib().go_healpix_pix2vec_nest
    f.argtypes = [C.c_long, C.c_long, C.c_void_p]
    for i in range(n):
        f(nside, i, out[i].ctypes.data)
    return out


def orthogonal_rays_z(n_side, mins4, maxs4):
    rays = np.empty(n_side * n_side, RAY_DTYPE)
    area = C.c_float(0)
    m = np.asarray(mins4, np.float32); M = np.asarray(maxs4, np.float32)
    lib().go_orthogonal_rays_z(C.c_int(n_side), _p(m), _p(M), _p(rays), C.byref(area))
    return rays, area.value


def ray_dir_keys(rays):
    rays = _rays(rays)
    f = lib().go_ray_dir_morton_key
    f.restype = C.c_uint32
    f.argtypes = [C.c_void_p]
    return np.array([f(rays[i:i + 1].ctypes.data) for i in range(len(rays))], np.uint32)


# -- triangles (tests/profile_trace_triangle) -------------------------------------------

def _tris(t):
    t = np.ascontiguousarray(t, np.float32)
    assert t.ndim == 2 and t.shape[1] == 9
    return t


def tri_centroid_bounds(tris):
    tris = _tris(tris)
    b = np.empty(3, np.float32); t = np.empty(3, np.float32)
    lib().go_tri_centroid_bounds(_p(tris), C.c_size_t(len(tris)), _p(b), _p(t))
    return b, t


def morton_keys30_tri(tris, bot, top):
    tris = _tris(tris)
    keys = np.empty(len(tris), np.uint32)
    b = np.asarray(bot, np.float32); t = np.asarray(top, np.float32)
    lib().go_morton_keys30_tri(_p(tris), C.c_size_t(len(tris)), _p(b), _p(t), _p(keys))
    return keys


def brute_closest_tri(rays, tris):
    rays = _rays(rays); tris = _tris(tris)
    out = np.empty(len(rays), np.int32); t = np.empty(len(rays), np.float32)
    lib().go_brute_closest_tri(_p(rays), C.c_size_t(len(rays)), _p(tris), C.c_size_t(len(tris)),
                               _p(out), _p(t))
    return out, t


def pinhole_rays(res_x, res_y, cam, look_at, up, fovy, length):
    rays = np.empty(res_x * res_y, RAY_DTYPE)
    f = lambda v: np.asarray(v, np.float32)
    a, b, c = f(cam), f(look_at), f(up)
    lib().go_pinhole_rays(C.c_int(res_x), C.c_int(res_y), _p(a), _p(b), _p(c), C.c_float(fovy),
                          C.c_float(length), _p(rays))
    return rays


def orthographic_projection_rays(res_x, res_y, cam, look_at, up, vertical_extent, length):
    rays = np.empty(res_x * res_y, RAY_DTYPE)
    f = lambda v: np.asarray(v, np.float32)
    a, b, c = f(cam), f(look_at), f(up)
    lib().go_orthographic_projection_rays(C.c_int(res_x), C.c_int(res_y), _p(a), _p(b), _p(c),
                                          C.c_float(vertical_extent), C.c_float(length), _p(rays))
    return rays


def one_to_many_rays(origin, points):
    """Unsorted rays origin -> points[i]; points [n, k>=3] float32 or float64."""
    points = np.ascontiguousarray(points)
    assert points.dtype in (np.float32, np.float64) and points.ndim == 2
    rays = np.empty(len(points), RAY_DTYPE)
    lib().go_one_to_many_rays(_p(points), C.c_int(int(points.dtype == np.float64)),
                              C.c_int(points.shape[1]), C.c_size_t(len(points)),
                              C.c_float(origin[0]), C.c_float(origin[1]), C.c_float(origin[2]),
                              _p(rays))
    return rays


# -- double4 spheres (Real4 = double4, Real = double) ------------------------------------------

def deltas_euclid_d4(prims):
    prims = np.ascontiguousarray(prims, np.float64)
    out = np.empty(len(prims) + 1, np.float32)
    lib().go_deltas_euclid_d4(_p(prims), C.c_size_t(len(prims)), _p(out))
    return out


def brute_hitcounts_d4(rays, prims):
    rays = _rays(rays); prims = np.ascontiguousarray(prims, np.float64)
    out = np.empty(len(rays), np.int32)
    lib().go_brute_hitcounts_d4(_p(rays), C.c_size_t(len(rays)), _p(prims), C.c_size_t(len(prims)), _p(out))
    return out


def brute_cumulative_d4(rays, prims, blocks=SUM_BLOCKS):
    """blocks > 1 (default): the class-ordered double sum (this implementation's stated result);
    blocks = 1: the reference's single running double sum."""
    rays = _rays(rays); prims = np.ascontiguousarray(prims, np.float64)
    out = np.empty(len(rays), np.float64)
    lib().go_brute_cumulative_d4(_p(rays), C.c_size_t(len(rays)), _p(prims), C.c_size_t(len(prims)), _p(out),
                                 C.c_int(blocks))
    return out

