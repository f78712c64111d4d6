// grace.h -- host-only C++ mirror of the grace:: header API for the BVH-build + SPH
// ray-traversal hot path, forwarding to the C ABI of libgrace_hip.so (include/grace_hip.h).
//
// Same names, argument meaning and error behaviour as the reference headers it replaces
// (paths relative to the reference root):
//   include/grace/ray.h, types.h                     -> grace::Ray, uinteger32/64
//   include/grace/cuda/nodes.h                       -> grace::Tree
//   include/grace/cuda/build_sph.cuh                 -> morton_keys*_sph, *_deltas_sph, ALBVH_sph
//   include/grace/cuda/trace_sph.cuh                 -> trace_hitcounts_sph, trace_cumulative_sph,
//                                                       trace_sph
//   include/grace/cuda/scan.cuh                      -> exclusive_segmented_scan
//   tests/helper/tree.cuh, tests/helper/rays.cuh     -> build_tree, orthogonal_rays_z
//   (no reference symbol; named by the task)         -> project_sph
//
// The reference's boundary type is thrust::device_vector; this mirror is HIP-free (plain
// g++ compiles it), so it ships grace::device_vector<T>, a minimal owning device array with
// the subset of the thrust interface the reference's call sites use (size, resize, data,
// assignment from / copy to std::vector).  Errors: a bad argument throws
// std::invalid_argument exactly where the reference does; a GPU API failure prints the
// message and exit()s like GRACE_CUDA_CHECK (include/grace/error.h:40-56).
#pragma once

#ifdef GRACE_DROPIN_HEADERS_INCLUDED
#error "grace/grace.h (HIP-free mirror) and the grace/cuda/*.cuh drop-in headers define the same names: include one set only"
#endif
#define GRACE_HIP_FREE_MIRROR_INCLUDED 1

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "grace_hip.h"
#include "grace/generic/morton.h"   // host-callable grace::morton_key, detail::space_by_two_* (generic/morton.h:14-55)
#include "grace/ray.h"   // grace::Ray (include/grace/ray.h:5-10)

namespace grace {

typedef uint32_t uinteger32; // include/grace/types.h:29-32
typedef uint64_t uinteger64;

struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int4 { int x, y, z, w; };
struct double3 { double x, y, z; };
struct double4 { double x, y, z, w; };

// include/grace/types.h:36-51
enum Octants { PPP = 7, PPM = 6, PMP = 5, PMM = 4, MPP = 3, MPM = 2, MMP = 1, MMM = 0 };
enum RaySortType { NoSort, DirectionSort, EndPointSort };

inline float3 make_float3(float x, float y, float z) { float3 v = { x, y, z }; return v; }
inline float4 make_float4(float x, float y, float z, float w) { float4 v = { x, y, z, w }; return v; }

namespace detail {

inline void check(grace_status s)
{
    if (s == GRACE_OK) return;
    if (s == GRACE_INVALID_ARGUMENT) throw std::invalid_argument(grace_last_error());
    // include/grace/error.h:40-56: print and exit with the error code.
    std::fprintf(stderr, "**** GRACE HIP Error ****\n%s\n", grace_last_error());
    std::exit(int(s));
}

} // namespace detail

// Minimal stand-in for thrust::device_vector<T> (device memory owned through the C ABI).
template <typename T>
class device_vector {
public:
    device_vector() : ptr_(nullptr), size_(0), capacity_(0) {}
    explicit device_vector(size_t n) : ptr_(nullptr), size_(0), capacity_(0)
    {
        resize(n);
        if (n) detail::check(grace_memset(ptr_, 0, n * sizeof(T), nullptr)); // value-init
    }
    device_vector(const std::vector<T>& h) : ptr_(nullptr), size_(0), capacity_(0) { *this = h; }
    device_vector(const device_vector& o) : ptr_(nullptr), size_(0), capacity_(0)
    {
        resize(o.size_);
        detail::check(grace_memcpy_dtod(ptr_, o.ptr_, size_ * sizeof(T), nullptr));
    }
    ~device_vector() { grace_device_free(ptr_); }

    device_vector& operator=(const std::vector<T>& h)
    {
        resize(h.size());
        detail::check(grace_memcpy_htod(ptr_, h.data(), h.size() * sizeof(T), nullptr));
        return *this;
    }
    device_vector& operator=(const device_vector& o)
    {
        if (this != &o) {
            resize(o.size_);
            detail::check(grace_memcpy_dtod(ptr_, o.ptr_, size_ * sizeof(T), nullptr));
        }
        return *this;
    }

    // Keeps the old contents (thrust semantics); new elements are unspecified.
    void resize(size_t n)
    {
        if (n > capacity_) {
            void* p = nullptr;
            detail::check(grace_device_malloc(&p, n * sizeof(T)));
            if (size_) detail::check(grace_memcpy_dtod(p, ptr_, size_ * sizeof(T), nullptr));
            detail::check(grace_stream_synchronize(nullptr));
            grace_device_free(ptr_);
            ptr_ = static_cast<T*>(p);
            capacity_ = n;
        }
        size_ = n;
    }
    size_t size() const { return size_; }
    T* data() { return ptr_; }
    const T* data() const { return ptr_; }

    std::vector<T> to_host() const
    {
        std::vector<T> h(size_);
        detail::check(grace_memcpy_dtoh(h.data(), ptr_, size_ * sizeof(T), nullptr));
        return h;
    }
    T at_host(size_t i) const
    {
        T v;
        detail::check(grace_memcpy_dtoh(&v, ptr_ + i, sizeof(T), nullptr));
        return v;
    }

private:
    T* ptr_;
    size_t size_, capacity_;
};

// include/grace/cuda/nodes.h:14-58.  nodes holds 4 int4 per node, leaves one int4 per leaf;
// both are allocated for N leaves and shrunk by the build (albvh.cuh:842-845).
class Tree {
public:
    device_vector<int4> nodes;
    device_vector<int4> leaves;
    int* root_index_ptr;
    int max_per_leaf;

    Tree(size_t N_leaves, int max_per_leaf_ = 1)
        : nodes(4 * (N_leaves - 1)), leaves(N_leaves), root_index_ptr(nullptr),
          max_per_leaf(max_per_leaf_)
    {
        void* p = nullptr;
        detail::check(grace_device_malloc(&p, sizeof(int)));
        root_index_ptr = static_cast<int*>(p);
    }
    ~Tree() { grace_device_free(root_index_ptr); }

private:
    Tree(const Tree&);
    Tree& operator=(const Tree&);
};

// ---- build: include/grace/cuda/build_sph.cuh -------------------------------------------

// build_sph.cuh:27-35 (30-bit keys)
inline void morton_keys_sph(const device_vector<float4>& d_spheres, const float3 bot,
                            const float3 top, device_vector<uinteger32>& d_keys)
{
    const float b[3] = { bot.x, bot.y, bot.z }, t[3] = { top.x, top.y, top.z };
    detail::check(grace_morton_keys30_f4(&d_spheres.data()->x, d_spheres.size(), b, t,
                                         d_keys.data(), nullptr));
}

// build_sph.cuh:27-35 (63-bit keys)
inline void morton_keys_sph(const device_vector<float4>& d_spheres, const float3 bot,
                            const float3 top, device_vector<uinteger64>& d_keys)
{
    const float b[3] = { bot.x, bot.y, bot.z }, t[3] = { top.x, top.y, top.z };
    detail::check(grace_morton_keys63_f4(&d_spheres.data()->x, d_spheres.size(), b, t,
                                         d_keys.data(), nullptr));
}

// build_sph.cuh:19-25: bounds from the centroids (kernels/morton.cuh:139-174)
template <typename KeyType>
inline void morton_keys_sph(const device_vector<float4>& d_spheres, device_vector<KeyType>& d_keys)
{
    float b[3], t[3];
    detail::check(grace_centroid_bounds_f4(&d_spheres.data()->x, d_spheres.size(), b, t, nullptr));
    morton_keys_sph(d_spheres, make_float3(b[0], b[1], b[2]), make_float3(t[0], t[1], t[2]),
                    d_keys);
}

// build_sph.cuh:50-58
inline void morton_keys30_sort_sph(device_vector<float4>& d_spheres, const float3 bot,
                                   const float3 top)
{
    device_vector<uinteger32> d_keys;
    d_keys.resize(d_spheres.size());
    morton_keys_sph(d_spheres, bot, top, d_keys);
    detail::check(grace_sort_pairs_u32(d_keys.data(), d_spheres.data(), d_spheres.size(),
                                       sizeof(float4), 0, 30, nullptr, nullptr));
}

// build_sph.cuh:41-47
inline void morton_keys30_sort_sph(device_vector<float4>& d_spheres)
{
    device_vector<uinteger32> d_keys;
    d_keys.resize(d_spheres.size());
    morton_keys_sph(d_spheres, d_keys);
    detail::check(grace_sort_pairs_u32(d_keys.data(), d_spheres.data(), d_spheres.size(),
                                       sizeof(float4), 0, 30, nullptr, nullptr));
}

// build_sph.cuh:74-82
inline void morton_keys63_sort_sph(device_vector<float4>& d_spheres, const float3 bot,
                                   const float3 top)
{
    device_vector<uinteger64> d_keys;
    d_keys.resize(d_spheres.size());
    morton_keys_sph(d_spheres, bot, top, d_keys);
    detail::check(grace_sort_pairs_u64(d_keys.data(), d_spheres.data(), d_spheres.size(),
                                       sizeof(float4), 0, 63, nullptr, nullptr));
}

// build_sph.cuh:87-94
inline void euclidean_deltas_sph(const device_vector<float4>& d_spheres,
                                 device_vector<float>& d_deltas)
{
    detail::check(grace_deltas_euclid_f4(&d_spheres.data()->x, d_spheres.size(), d_deltas.data(),
                                         nullptr));
}

// build_sph.cuh:98-105
inline void surface_area_deltas_sph(const device_vector<float4>& d_spheres,
                                    device_vector<float>& d_deltas)
{
    detail::check(grace_deltas_area_f4(&d_spheres.data()->x, d_spheres.size(), d_deltas.data(),
                                       nullptr));
}

// build_sph.cuh:109-114
inline void XOR_deltas_sph(const device_vector<uinteger32>& d_keys,
                           device_vector<uinteger32>& d_deltas)
{
    detail::check(grace_deltas_xor_u32(d_keys.data(), d_keys.size(), d_deltas.data(), nullptr));
}

inline void XOR_deltas_sph(const device_vector<uinteger64>& d_keys,
                           device_vector<uinteger64>& d_deltas)
{
    detail::check(grace_deltas_xor_u64(d_keys.data(), d_keys.size(), d_deltas.data(), nullptr));
}

// build_sph.cuh:118-124 -> build_ALBVH (kernels/albvh.cuh:986-1021)
inline void ALBVH_sph(const device_vector<float4>& d_spheres, const device_vector<float>& d_deltas,
                      Tree& d_tree)
{
    size_t n_leaves = 0;
    detail::check(grace_albvh_build_f4(&d_spheres.data()->x, d_spheres.size(), d_deltas.data(),
                                       d_tree.max_per_leaf, &d_tree.nodes.data()->x,
                                       &d_tree.leaves.data()->x, d_tree.root_index_ptr, &n_leaves,
                                       nullptr));
    d_tree.nodes.resize(4 * (n_leaves - 1));
    d_tree.leaves.resize(n_leaves);
}

inline void ALBVH_sph(const device_vector<float4>& d_spheres,
                      const device_vector<uinteger32>& d_deltas, Tree& d_tree)
{
    size_t n_leaves = 0;
    detail::check(grace_albvh_build_f4_u32(&d_spheres.data()->x, d_spheres.size(), d_deltas.data(),
                                           d_tree.max_per_leaf, &d_tree.nodes.data()->x,
                                           &d_tree.leaves.data()->x, d_tree.root_index_ptr,
                                           &n_leaves, nullptr));
    d_tree.nodes.resize(4 * (n_leaves - 1));
    d_tree.leaves.resize(n_leaves);
}

// ---- trace: include/grace/cuda/trace_sph.cuh --------------------------------------------

namespace detail {

inline void check_ray_count(size_t n_rays)
{
    // include/grace/cuda/kernels/bintree_trace.cuh:231-238
    if (n_rays % 32 != 0)
        throw std::invalid_argument("Number of rays must be a multiple of the warp size (32).");
}

} // namespace detail

// trace_sph.cuh:58-80
inline void trace_hitcounts_sph(const device_vector<Ray>& d_rays,
                                const device_vector<float4>& d_spheres, const Tree& d_tree,
                                device_vector<int>& d_hit_counts)
{
    detail::check_ray_count(d_rays.size());
    detail::check(grace_trace_hitcounts_f4(d_rays.data(), d_rays.size(), &d_spheres.data()->x,
                                           d_spheres.size(), &d_tree.nodes.data()->x,
                                           d_tree.leaves.size() - 1, &d_tree.leaves.data()->x,
                                           d_tree.root_index_ptr, d_hit_counts.data(), nullptr));
    detail::check(grace_trace_status(nullptr));
}

// trace_sph.cuh:82-110
inline void trace_cumulative_sph(const device_vector<Ray>& d_rays,
                                 const device_vector<float4>& d_spheres, const Tree& d_tree,
                                 device_vector<float>& d_cumulated)
{
    detail::check_ray_count(d_rays.size());
    detail::check(grace_trace_cumulative_f4(d_rays.data(), d_rays.size(), &d_spheres.data()->x,
                                            d_spheres.size(), &d_tree.nodes.data()->x,
                                            d_tree.leaves.size() - 1, &d_tree.leaves.data()->x,
                                            d_tree.root_index_ptr, d_cumulated.data(), nullptr));
    detail::check(grace_trace_status(nullptr));
}

// trace_sph.cuh:112-168
inline void trace_sph(const device_vector<Ray>& d_rays, const device_vector<float4>& d_spheres,
                      const Tree& d_tree, device_vector<int>& d_ray_offsets,
                      device_vector<int>& d_hit_indices, device_vector<float>& d_hit_integrals,
                      device_vector<float>& d_hit_distances)
{
    // (the hit-count pass made for a per-hit trace: the library keeps what pass 2 can reuse)
    detail::check_ray_count(d_rays.size());
    detail::check(grace_trace_hitcounts_keep_f4(d_rays.data(), d_rays.size(), &d_spheres.data()->x,
                                                d_spheres.size(), &d_tree.nodes.data()->x,
                                                d_tree.leaves.size() - 1, &d_tree.leaves.data()->x,
                                                d_tree.root_index_ptr, d_ray_offsets.data(), nullptr));
    long long total = 0;
    detail::check(grace_scan_exclusive_i32(d_ray_offsets.data(), d_ray_offsets.size(),
                                           d_ray_offsets.data(), &total, nullptr));
    if (total > 2147483647LL)   // int offsets cannot address more (the reference's int scan would wrap)
        throw std::invalid_argument("trace_sph: more than INT_MAX hits; trace fewer rays per call.");
    d_hit_integrals.resize(size_t(total));
    d_hit_indices.resize(size_t(total));
    d_hit_distances.resize(size_t(total));
    if (total == 0) return;   // no ray hits anything: empty vectors, as thrust's resize(0)
    detail::check(grace_trace_hits_f4(d_rays.data(), d_rays.size(), &d_spheres.data()->x,
                                      d_spheres.size(), &d_tree.nodes.data()->x,
                                      d_tree.leaves.size() - 1, &d_tree.leaves.data()->x,
                                      d_tree.root_index_ptr, d_ray_offsets.data(),
                                      d_hit_indices.data(), d_hit_integrals.data(),
                                      d_hit_distances.data(), nullptr));
    detail::check(grace_trace_status(nullptr));
}

// trace_sph.cuh:171-241
inline void trace_with_sentinels_sph(const device_vector<Ray>& d_rays,
                                     const device_vector<float4>& d_spheres, const Tree& d_tree,
                                     device_vector<int>& d_ray_offsets,
                                     device_vector<int>& d_hit_indices, const int index_sentinel,
                                     device_vector<float>& d_hit_integrals,
                                     const float integral_sentinel,
                                     device_vector<float>& d_hit_distances,
                                     const float distance_sentinel)
{
    const size_t n_rays = d_rays.size();
    trace_hitcounts_sph(d_rays, d_spheres, d_tree, d_ray_offsets);
    long long total = 0;
    detail::check(grace_scan_exclusive_i32(d_ray_offsets.data(), n_rays, d_ray_offsets.data(),
                                           &total, nullptr));
    if (total + (long long)n_rays > 2147483647LL)
        throw std::invalid_argument("trace_with_sentinels_sph: more than INT_MAX output slots; "
                                    "trace fewer rays per call.");
    const size_t allocate_size = size_t(total) + n_rays;
    detail::check(grace_add_iota_i32(d_ray_offsets.data(), n_rays, nullptr));
    d_hit_indices.resize(allocate_size);
    d_hit_integrals.resize(allocate_size);
    d_hit_distances.resize(allocate_size);
    uint32_t ib, db;
    std::memcpy(&ib, &integral_sentinel, 4);
    std::memcpy(&db, &distance_sentinel, 4);
    detail::check(grace_fill_u32(d_hit_indices.data(), allocate_size, uint32_t(index_sentinel), nullptr));
    detail::check(grace_fill_u32(d_hit_integrals.data(), allocate_size, ib, nullptr));
    detail::check(grace_fill_u32(d_hit_distances.data(), allocate_size, db, nullptr));
    detail::check(grace_trace_hits_f4(d_rays.data(), n_rays, &d_spheres.data()->x, d_spheres.size(),
                                      &d_tree.nodes.data()->x, d_tree.leaves.size() - 1,
                                      &d_tree.leaves.data()->x, d_tree.root_index_ptr,
                                      d_ray_offsets.data(), d_hit_indices.data(),
                                      d_hit_integrals.data(), d_hit_distances.data(), nullptr));
    detail::check(grace_trace_status(nullptr));
}

// ---- scan: include/grace/cuda/scan.cuh:15-37 --------------------------------------------
inline void exclusive_segmented_scan(const device_vector<int>& d_segment_offsets,
                                     device_vector<float>& d_data, device_vector<float>& d_results)
{
    detail::check(grace_segscan_exclusive_f32(d_segment_offsets.data(), d_segment_offsets.size(),
                                              d_data.data(), d_data.size(), d_results.data(),
                                              nullptr));
}

inline void exclusive_segmented_scan(const device_vector<int>& d_segment_offsets,
                                     device_vector<double>& d_data,
                                     device_vector<double>& d_results)
{
    detail::check(grace_segscan_exclusive_f64(d_segment_offsets.data(), d_segment_offsets.size(),
                                              d_data.data(), d_data.size(), d_results.data(),
                                              nullptr));
}

// include/grace/cuda/sort.cuh:100-131
inline void sort_by_distance(device_vector<float>& d_hit_distances,
                             const device_vector<int>& d_ray_offsets,
                             device_vector<int>& d_hit_indices, device_vector<float>& d_hit_data)
{
    detail::check(grace_sort_by_distance_f32(d_hit_distances.data(), d_ray_offsets.data(),
                                             d_ray_offsets.size(), d_hit_distances.size(),
                                             d_hit_indices.data(), d_hit_data.data(), nullptr));
}

// ---- ray generators, include/grace/cuda/gen_rays.cuh (vector overloads: d_rays is grown when
// too small, never shrunk).  Random generators use this library's own counter-based streams;
// the reference's cuRAND streams are device-specific by its own account (gen_rays.cuh:21-24).
namespace detail {
template <typename T> struct point_traits;   // PointType: any of the four below
template <> struct point_traits<float3> { enum { is_double = 0, elems = 3 }; };
template <> struct point_traits<float4> { enum { is_double = 0, elems = 4 }; };
template <> struct point_traits<double3> { enum { is_double = 1, elems = 3 }; };
template <> struct point_traits<double4> { enum { is_double = 1, elems = 4 }; };
} // namespace detail

// gen_rays.cuh:25-60
inline void uniform_random_rays(device_vector<Ray>& d_rays, const float ox, const float oy,
                                const float oz, const float length,
                                const unsigned long long seed = 1234)
{
    detail::check(grace_rays_isotropic(d_rays.size(), ox, oy, oz, length, seed, d_rays.data(),
                                       nullptr));
}

// gen_rays.cuh:62-97
inline void uniform_random_rays_single_octant(device_vector<Ray>& d_rays, const float ox,
                                              const float oy, const float oz, const float length,
                                              const enum Octants octant = PPP,
                                              const unsigned long long seed = 1234)
{
    detail::check(grace_rays_isotropic_octant(d_rays.size(), ox, oy, oz, length, int(octant), seed,
                                              d_rays.data(), nullptr));
}

// gen_rays.cuh:161-208 (bounds known) -- end-point sort.
template <typename PointType>
inline void one_to_many_rays(device_vector<Ray>& d_rays, const float ox, const float oy,
                             const float oz, const device_vector<PointType>& d_points,
                             const float3 AABB_bot, const float3 AABB_top)
{
    if (d_rays.size() < d_points.size()) d_rays.resize(d_points.size());
    detail::check(grace_rays_one_to_many(d_points.size(), ox, oy, oz, d_points.data(),
                                         detail::point_traits<PointType>::is_double,
                                         detail::point_traits<PointType>::elems, int(EndPointSort),
                                         &AABB_bot.x, &AABB_top.x, d_rays.data(), nullptr));
}

// gen_rays.cuh:99-159.  EndPointSort without bounds computes them from the points (the
// reference passes AABB_bot for both corners there, gen_rays.cuh:121-122: not reproduced).
template <typename PointType>
inline void one_to_many_rays(device_vector<Ray>& d_rays, const float ox, const float oy,
                             const float oz, const device_vector<PointType>& d_points,
                             const enum RaySortType sort_type = DirectionSort)
{
    if (sort_type != NoSort && sort_type != DirectionSort && sort_type != EndPointSort)
        throw std::invalid_argument("Ray sort type not recognized");
    if (sort_type == EndPointSort) {
        const std::vector<PointType> h = d_points.to_host();
        if (h.empty()) throw std::invalid_argument("one_to_many_rays: no points");
        float3 lo = make_float3(float(h[0].x), float(h[0].y), float(h[0].z)), hi = lo;
        for (size_t i = 1; i < h.size(); ++i) {
            const float x = float(h[i].x), y = float(h[i].y), z = float(h[i].z);
            lo.x = x < lo.x ? x : lo.x; hi.x = x > hi.x ? x : hi.x;
            lo.y = y < lo.y ? y : lo.y; hi.y = y > hi.y ? y : hi.y;
            lo.z = z < lo.z ? z : lo.z; hi.z = z > hi.z ? z : hi.z;
        }
        one_to_many_rays(d_rays, ox, oy, oz, d_points, lo, hi);
        return;
    }
    if (d_rays.size() < d_points.size()) d_rays.resize(d_points.size());
    detail::check(grace_rays_one_to_many(d_points.size(), ox, oy, oz, d_points.data(),
                                         detail::point_traits<PointType>::is_double,
                                         detail::point_traits<PointType>::elems, int(sort_type),
                                         nullptr, nullptr, d_rays.data(), nullptr));
}

// gen_rays.cuh:210-262
inline void plane_parallel_random_rays(device_vector<Ray>& d_rays, const int width, const int height,
                                       const float3 base, const float3 w, const float3 h,
                                       const float length, const unsigned long long seed = 1234)
{
    const size_t n = size_t(width) * height;
    if (d_rays.size() < n) d_rays.resize(n);
    detail::check(grace_rays_plane_parallel_random(width, height, &base.x, &w.x, &h.x, length, seed,
                                                   d_rays.data(), nullptr));
}

// gen_rays.cuh:264-329
inline void orthographic_projection_rays(device_vector<Ray>& d_rays, const int resolution_x,
                                         const int resolution_y, const float3 camera_position,
                                         const float3 look_at, const float3 view_up,
                                         const float vertical_extent, const float length)
{
    const size_t n = size_t(resolution_x) * resolution_y;
    if (d_rays.size() < n) d_rays.resize(n);
    detail::check(grace_rays_orthographic_projection(resolution_x, resolution_y, &camera_position.x,
                                                     &look_at.x, &view_up.x, vertical_extent, length,
                                                     d_rays.data(), nullptr));
}

// gen_rays.cuh:331-399
inline void pinhole_camera_rays(device_vector<Ray>& d_rays, const int resolution_x,
                                const int resolution_y, const float3 camera_position,
                                const float3 look_at, const float3 view_up, const float FOVy,
                                const float length)
{
    const size_t n = size_t(resolution_x) * resolution_y;
    if (d_rays.size() < n) d_rays.resize(n);
    detail::check(grace_rays_pinhole(resolution_x, resolution_y, &camera_position.x, &look_at.x,
                                     &view_up.x, FOVy, length, d_rays.data(), nullptr));
}

// ---- double4 particles (build_sph.cuh:16-82 with Real4 = double4): keys from the co-ordinates
// narrowed to float (CentroidSphere), 32-byte records moved by the sort.
inline void morton_keys_sph(const device_vector<double4>& d_spheres, const float3 bot,
                            const float3 top, device_vector<uinteger32>& d_keys)
{
    detail::check(grace_morton_keys30_points(d_spheres.data(), d_spheres.size(), 1, 4, &bot.x, &top.x,
                                             d_keys.data(), nullptr));
}
inline void morton_keys_sph(const device_vector<double4>& d_spheres, const float3 bot,
                            const float3 top, device_vector<uinteger64>& d_keys)
{
    detail::check(grace_morton_keys63_points(d_spheres.data(), d_spheres.size(), 1, 4, &bot.x, &top.x,
                                             d_keys.data(), nullptr));
}
inline void morton_keys30_sort_sph(device_vector<double4>& d_spheres, const float3 bot,
                                   const float3 top)
{
    device_vector<uinteger32> d_keys(d_spheres.size());
    morton_keys_sph(d_spheres, bot, top, d_keys);
    detail::check(grace_sort_pairs_u32(d_keys.data(), d_spheres.data(), d_spheres.size(), 32, 0, 30,
                                       nullptr, nullptr));
}
inline void morton_keys63_sort_sph(device_vector<double4>& d_spheres, const float3 bot,
                                   const float3 top)
{
    device_vector<uinteger64> d_keys(d_spheres.size());
    morton_keys_sph(d_spheres, bot, top, d_keys);
    detail::check(grace_sort_pairs_u64(d_keys.data(), d_spheres.data(), d_spheres.size(), 32, 0, 63,
                                       nullptr, nullptr));
}

// build_sph.cuh:84-93 with Real4 = double4.  The functor returns float whatever Real is
// (generic/functors/albvh.h:44-74), so the deltas are kept as float here (the reference stores
// the same values widened into a vector<double>).
inline void euclidean_deltas_sph(const device_vector<double4>& d_spheres,
                                 device_vector<float>& d_deltas)
{
    detail::check(grace_deltas_euclid_d4(&d_spheres.data()->x, d_spheres.size(), d_deltas.data(),
                                         nullptr));
}

// build_sph.cuh:116-126 with Real4 = double4
inline void ALBVH_sph(const device_vector<double4>& d_spheres, const device_vector<float>& d_deltas,
                      Tree& d_tree)
{
    size_t n_leaves = 0;
    detail::check(grace_albvh_build_d4(&d_spheres.data()->x, d_spheres.size(), d_deltas.data(),
                                       d_tree.max_per_leaf, &d_tree.nodes.data()->x,
                                       &d_tree.leaves.data()->x, d_tree.root_index_ptr, &n_leaves,
                                       nullptr));
    d_tree.nodes.resize(4 * (n_leaves - 1));
    d_tree.leaves.resize(n_leaves);
}

// trace_sph.cuh:57-110 with Real4 = double4, Real = double
inline void trace_hitcounts_sph(const device_vector<Ray>& d_rays,
                                const device_vector<double4>& d_spheres, const Tree& d_tree,
                                device_vector<int>& d_hit_counts)
{
    detail::check_ray_count(d_rays.size());
    detail::check(grace_trace_hitcounts_d4(d_rays.data(), d_rays.size(), &d_spheres.data()->x,
                                           d_spheres.size(), &d_tree.nodes.data()->x,
                                           d_tree.leaves.size() - 1, &d_tree.leaves.data()->x,
                                           d_tree.root_index_ptr, d_hit_counts.data(), nullptr));
    detail::check(grace_trace_status_d4(nullptr));
}

inline void trace_cumulative_sph(const device_vector<Ray>& d_rays,
                                 const device_vector<double4>& d_spheres, const Tree& d_tree,
                                 device_vector<double>& d_cumulated)
{
    detail::check_ray_count(d_rays.size());
    detail::check(grace_trace_cumulative_d4(d_rays.data(), d_rays.size(), &d_spheres.data()->x,
                                            d_spheres.size(), &d_tree.nodes.data()->x,
                                            d_tree.leaves.size() - 1, &d_tree.leaves.data()->x,
                                            d_tree.root_index_ptr, d_cumulated.data(), nullptr));
    detail::check(grace_trace_status_d4(nullptr));
}

// Extensions (not in the reference; see grace_hip.h): what every trace call otherwise recomputes
// from its arguments -- the scene's pre-pass records, the ray coherence order -- computed once for
// inputs that are traced repeatedly.  Results never depend on it.
// The returned handle pins the cached records while it lives; they are validated against the
// arrays' current contents before every use (see "Cached trace records" in grace_hip.h).
class PreparedTrace
{
public:
    PreparedTrace() : scene_(false), rays_(false) {}
    PreparedTrace(PreparedTrace&& o) : scene_(o.scene_), rays_(o.rays_) { o.scene_ = o.rays_ = false; }
    PreparedTrace& operator=(PreparedTrace&& o)
    {
        if (this != &o) { release(); scene_ = o.scene_; rays_ = o.rays_; o.scene_ = o.rays_ = false; }
        return *this;
    }
    ~PreparedTrace() { release(); }
    void release()
    {
        if (scene_) detail::check(grace_trace_release());
        if (rays_) detail::check(grace_trace_release_rays());
        scene_ = rays_ = false;
    }

private:
    PreparedTrace(const PreparedTrace&);
    PreparedTrace& operator=(const PreparedTrace&);
    bool scene_, rays_;
    friend PreparedTrace prepare_trace_sph(const device_vector<float4>&, const Tree&);
    friend PreparedTrace prepare_trace_rays(const device_vector<Ray>&);
};

__attribute__((warn_unused_result))
inline PreparedTrace prepare_trace_sph(const device_vector<float4>& d_spheres, const Tree& d_tree)
{
    detail::check(grace_trace_prepare_f4(&d_spheres.data()->x, d_spheres.size(), &d_tree.nodes.data()->x,
                                         d_tree.leaves.size() - 1, &d_tree.leaves.data()->x, nullptr));
    PreparedTrace h;
    h.scene_ = true;
    return h;
}

__attribute__((warn_unused_result))
inline PreparedTrace prepare_trace_rays(const device_vector<Ray>& d_rays)
{
    detail::check(grace_trace_prepare_rays(d_rays.data(), d_rays.size(), nullptr));
    PreparedTrace h;
    h.rays_ = true;
    return h;
}

inline void release_prepared_trace()
{
    detail::check(grace_trace_release());
    detail::check(grace_trace_release_rays());
}

// util/extrema.cuh min_vec4 / max_vec4 as used by tests/project_gadget/project_gadget.cu:66-68
inline void min_max_vec4(const device_vector<float4>& d_v, float4* mins, float4* maxs)
{
    detail::check(grace_minmax_f4(&d_v.data()->x, d_v.size(), &mins->x, &maxs->x, nullptr));
}

} // namespace grace

// ---- test helpers the task names as API (global namespace in the reference) -------------

// tests/helper/tree.cuh:30-43
inline void build_tree(grace::device_vector<grace::float4>& spheres, const grace::float4 low,
                       const grace::float4 high, grace::Tree& tree)
{
    grace::device_vector<float> deltas;
    deltas.resize(spheres.size() + 1);
    grace::morton_keys30_sort_sph(spheres, grace::make_float3(low.x, low.y, low.z),
                                  grace::make_float3(high.x, high.y, high.z));
    grace::euclidean_deltas_sph(spheres, deltas);
    grace::ALBVH_sph(spheres, deltas, tree);
}

// tests/helper/tree.cuh:15-25
inline void build_tree(grace::device_vector<grace::float4>& spheres, grace::Tree& tree)
{
    grace::device_vector<float> deltas;
    deltas.resize(spheres.size() + 1);
    grace::morton_keys30_sort_sph(spheres);
    grace::euclidean_deltas_sph(spheres, deltas);
    grace::ALBVH_sph(spheres, deltas, tree);
}

// tests/helper/rays.cuh:55-79
inline void orthogonal_rays_z(const size_t N_side, const grace::float4 mins,
                              const grace::float4 maxs, grace::device_vector<grace::Ray>& d_rays,
                              float* area = NULL)
{
    d_rays.resize(N_side * N_side);
    grace::detail::check(grace_rays_orthogonal_z(int(N_side), &mins.x, &maxs.x, d_rays.data(), area,
                                                 nullptr));
}

// The projection of tests/project_gadget/project_gadget.cu:58-81 as one call: bounds with
// w = 0, build_tree (sorts d_spheres), orthogonal_rays_z, trace_cumulative_sph.
inline void project_sph(grace::device_vector<grace::float4>& d_spheres, const size_t N_side,
                        const int max_per_leaf, grace::device_vector<float>& d_image)
{
    grace::float4 mins, maxs;
    grace::min_max_vec4(d_spheres, &mins, &maxs);
    mins.w = maxs.w = 0;
    grace::Tree tree(d_spheres.size(), max_per_leaf);
    build_tree(d_spheres, mins, maxs, tree);
    grace::device_vector<grace::Ray> rays;
    orthogonal_rays_z(N_side, mins, maxs, rays);
    d_image.resize(rays.size());
    grace::trace_cumulative_sph(rays, d_spheres, tree, d_image);
}
