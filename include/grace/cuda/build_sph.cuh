// grace/cuda/build_sph.cuh -- the SPH build API of the reference
// (include/grace/cuda/build_sph.cuh:16-124) with its template signatures over
// thrust::device_vector, every body a type dispatch onto libgrace_hip.so:
//
//   morton_keys_sph            -> grace_morton_keys{30,63}_f4[_d3] / _points[_d3]   (csrc/morton.hip)
//   morton_keys{30,63}_sort_sph-> the same + grace_sort_pairs_u32/u64               (csrc/sort.hip;
//                                 the reference calls thrust::sort_by_key: stable, in place)
//   euclidean / surface_area / XOR _deltas_sph -> grace_deltas_*                    (csrc/deltas.hip)
//   ALBVH_sph                  -> grace_albvh_build_*                               (csrc/albvh.hip)
//
// Real4 is float4 or double4, KeyType / XOR DeltaType uinteger32 or uinteger64, Real the scalar
// type of Real4 (as the reference requires, build_sph.cuh:84-86).  rocThrust is the container
// only: no Thrust algorithm runs on this path.
#pragma once

#include "grace/cuda/nodes.h"
#include "grace/detail/raw.h"
#include "grace/generic/meta.h"

namespace grace {

namespace detail {

// ---- Morton keys: (Real4, bounds precision, key type) -> entry point -----------------------
inline void keys_dispatch(const float4* s, size_t n, const float* b, const float* t, uinteger32* k)
{ GRACE_STATUS_CHECK(grace_morton_keys30_f4(reinterpret_cast<const float*>(s), n, b, t, k, NULL)); }
inline void keys_dispatch(const float4* s, size_t n, const float* b, const float* t, uinteger64* k)
{ GRACE_STATUS_CHECK(grace_morton_keys63_f4(reinterpret_cast<const float*>(s), n, b, t, k, NULL)); }
inline void keys_dispatch(const float4* s, size_t n, const double* b, const double* t, uinteger32* k)
{ GRACE_STATUS_CHECK(grace_morton_keys30_f4_d3(reinterpret_cast<const float*>(s), n, b, t, k, NULL)); }
inline void keys_dispatch(const float4* s, size_t n, const double* b, const double* t, uinteger64* k)
{ GRACE_STATUS_CHECK(grace_morton_keys63_f4_d3(reinterpret_cast<const float*>(s), n, b, t, k, NULL)); }
inline void keys_dispatch(const double4* s, size_t n, const float* b, const float* t, uinteger32* k)
{ GRACE_STATUS_CHECK(grace_morton_keys30_points(s, n, 1, 4, b, t, k, NULL)); }
inline void keys_dispatch(const double4* s, size_t n, const float* b, const float* t, uinteger64* k)
{ GRACE_STATUS_CHECK(grace_morton_keys63_points(s, n, 1, 4, b, t, k, NULL)); }
inline void keys_dispatch(const double4* s, size_t n, const double* b, const double* t, uinteger32* k)
{ GRACE_STATUS_CHECK(grace_morton_keys30_points_d3(s, n, 1, 4, b, t, k, NULL)); }
inline void keys_dispatch(const double4* s, size_t n, const double* b, const double* t, uinteger64* k)
{ GRACE_STATUS_CHECK(grace_morton_keys63_points_d3(s, n, 1, 4, b, t, k, NULL)); }

// Centroid bounds of the spheres (the bounds-free overloads: compute_centroids + min/max,
// kernels/morton.cuh:139-174); centroids are float3 for either Real4.
inline void centroid_bounds(const float4* s, size_t n, float* b, float* t)
{ GRACE_STATUS_CHECK(grace_centroid_bounds_f4(reinterpret_cast<const float*>(s), n, b, t, NULL)); }
inline void centroid_bounds(const double4* s, size_t n, float* b, float* t)
{ GRACE_STATUS_CHECK(grace_centroid_bounds_points(s, n, 1, 4, b, t, NULL)); }

// ---- stable sort of the spheres by key, in place --------------------------------------------
template <typename Real4>
inline void sort_spheres(thrust::device_vector<uinteger32>& k, thrust::device_vector<Real4>& s, int bits)
{ GRACE_STATUS_CHECK(grace_sort_pairs_u32(raw(k), raw(s), s.size(), int(sizeof(Real4)), 0, bits, NULL, NULL)); }
template <typename Real4>
inline void sort_spheres(thrust::device_vector<uinteger64>& k, thrust::device_vector<Real4>& s, int bits)
{ GRACE_STATUS_CHECK(grace_sort_pairs_u64(raw(k), raw(s), s.size(), int(sizeof(Real4)), 0, bits, NULL, NULL)); }

// ---- deltas ------------------------------------------------------------------------------------
inline void euclid_dispatch(const float4* s, size_t n, float* d)
{ GRACE_STATUS_CHECK(grace_deltas_euclid_f4(reinterpret_cast<const float*>(s), n, d, NULL)); }
inline void euclid_dispatch(const double4* s, size_t n, double* d)
{ GRACE_STATUS_CHECK(grace_deltas_euclid_d4_f64(reinterpret_cast<const double*>(s), n, d, NULL)); }
inline void euclid_dispatch(const double4* s, size_t n, float* d)
{ GRACE_STATUS_CHECK(grace_deltas_euclid_d4(reinterpret_cast<const double*>(s), n, d, NULL)); }
inline void area_dispatch(const float4* s, size_t n, float* d)
{ GRACE_STATUS_CHECK(grace_deltas_area_f4(reinterpret_cast<const float*>(s), n, d, NULL)); }
inline void area_dispatch(const double4* s, size_t n, double* d)
{ GRACE_STATUS_CHECK(grace_deltas_area_d4_f64(reinterpret_cast<const double*>(s), n, d, NULL)); }
inline void area_dispatch(const double4* s, size_t n, float* d)
{ GRACE_STATUS_CHECK(grace_deltas_area_d4(reinterpret_cast<const double*>(s), n, d, NULL)); }
inline void xor_dispatch(const uinteger32* k, size_t n, uinteger32* d)
{ GRACE_STATUS_CHECK(grace_deltas_xor_u32(k, n, d, NULL)); }
inline void xor_dispatch(const uinteger64* k, size_t n, uinteger64* d)
{ GRACE_STATUS_CHECK(grace_deltas_xor_u64(k, n, d, NULL)); }

// ---- ALBVH: (Real4, DeltaType) -> entry point ------------------------------------------------
#define GRACE_ALBVH_DISPATCH(PRIM_T, CAST_T, DELTA_T, FN)                                        \
    inline void albvh_dispatch(const PRIM_T* s, size_t n, const DELTA_T* d, int mpl, int* nodes, \
                               int* leaves, int* root, size_t* n_leaves)                         \
    { GRACE_STATUS_CHECK(FN(reinterpret_cast<const CAST_T*>(s), n, d, mpl, nodes, leaves, root,  \
                            n_leaves, NULL)); }
GRACE_ALBVH_DISPATCH(float4, float, float, grace_albvh_build_f4)
GRACE_ALBVH_DISPATCH(float4, float, double, grace_albvh_build_f4_f64)
GRACE_ALBVH_DISPATCH(float4, float, uinteger32, grace_albvh_build_f4_u32)
GRACE_ALBVH_DISPATCH(float4, float, uinteger64, grace_albvh_build_f4_u64)
GRACE_ALBVH_DISPATCH(double4, double, float, grace_albvh_build_d4)
GRACE_ALBVH_DISPATCH(double4, double, double, grace_albvh_build_d4_f64)
GRACE_ALBVH_DISPATCH(double4, double, uinteger32, grace_albvh_build_d4_u32)
GRACE_ALBVH_DISPATCH(double4, double, uinteger64, grace_albvh_build_d4_u64)
#undef GRACE_ALBVH_DISPATCH

// Bounds arrive as any type with .x/.y/.z; the arithmetic precision is that of Real3's
// components (kernels/morton.cuh:104-113).
template <typename Real3> struct bounds_scalar { typedef float type; };
template <> struct bounds_scalar<double3> { typedef double type; };
template <> struct bounds_scalar<double4> { typedef double type; };

} // namespace detail

// Real4 should be float4 or double4.
// KeyType should be grace::uinteger{32,64}.
template <typename Real4, typename KeyType>
GRACE_HOST void morton_keys_sph(
    const thrust::device_vector<Real4>& d_spheres,
    thrust::device_vector<KeyType>& d_keys)
{
    float bot[3], top[3];
    detail::centroid_bounds(detail::raw(d_spheres), d_spheres.size(), bot, top);
    detail::keys_dispatch(detail::raw(d_spheres), d_spheres.size(), bot, top, detail::raw(d_keys));
}

template <typename Real3, typename Real4, typename KeyType>
GRACE_HOST void morton_keys_sph(
    const thrust::device_vector<Real4>& d_spheres,
    const Real3 bot,
    const Real3 top,
    thrust::device_vector<KeyType>& d_keys)
{
    typedef typename detail::bounds_scalar<Real3>::type B;
    B b[3], t[3];
    detail::xyz(bot, b);
    detail::xyz(top, t);
    detail::keys_dispatch(detail::raw(d_spheres), d_spheres.size(), b, t, detail::raw(d_keys));
}

// Generates 30-bit Morton keys and sorts the spheres by them (stable; in place).
// Requires O(N) on-device temporary storage.
template <typename Real4>
GRACE_HOST void morton_keys30_sort_sph(
    thrust::device_vector<Real4>& d_spheres)
{
    thrust::device_vector<grace::uinteger32> d_keys(d_spheres.size());
    morton_keys_sph(d_spheres, d_keys);
    detail::sort_spheres(d_keys, d_spheres, 30);
}

template <typename Real3, typename Real4>
GRACE_HOST void morton_keys30_sort_sph(
    thrust::device_vector<Real4>& d_spheres,
    const Real3 bot,
    const Real3 top)
{
    thrust::device_vector<grace::uinteger32> d_keys(d_spheres.size());
    morton_keys_sph(d_spheres, bot, top, d_keys);
    detail::sort_spheres(d_keys, d_spheres, 30);
}

// Generates 63-bit Morton keys and sorts the spheres by them.
template <typename Real4>
GRACE_HOST void morton_keys63_sort_sph(
    thrust::device_vector<Real4>& d_spheres)
{
    thrust::device_vector<grace::uinteger64> d_keys(d_spheres.size());
    morton_keys_sph(d_spheres, d_keys);
    detail::sort_spheres(d_keys, d_spheres, 63);
}

template <typename Real3, typename Real4>
GRACE_HOST void morton_keys63_sort_sph(
    thrust::device_vector<Real4>& d_spheres,
    const Real3 bot,
    const Real3 top)
{
    thrust::device_vector<grace::uinteger64> d_keys(d_spheres.size());
    morton_keys_sph(d_spheres, bot, top, d_keys);
    detail::sort_spheres(d_keys, d_spheres, 63);
}

// Real4 should be float4 or double4.
// Real must be the float or double, respectively.
template <typename Real4, typename Real>
GRACE_HOST void euclidean_deltas_sph(
    const thrust::device_vector<Real4>& d_spheres,
    thrust::device_vector<Real>& d_deltas)
{
    GRACE_ASSERT(d_spheres.size() + 1 == d_deltas.size());
    detail::euclid_dispatch(detail::raw(d_spheres), d_spheres.size(), detail::raw(d_deltas));
}

template <typename Real4, typename Real>
GRACE_HOST void surface_area_deltas_sph(
    const thrust::device_vector<Real4>& d_spheres,
    thrust::device_vector<Real>& d_deltas)
{
    GRACE_ASSERT(d_spheres.size() + 1 == d_deltas.size());
    detail::area_dispatch(detail::raw(d_spheres), d_spheres.size(), detail::raw(d_deltas));
}

// KeyType should be grace::uinteger{32,64}; DeltaType the same type.
template <typename KeyType, typename DeltaType>
GRACE_HOST void XOR_deltas_sph(
    const thrust::device_vector<KeyType>& d_morton_keys,
    thrust::device_vector<DeltaType>& d_deltas)
{
    GRACE_ASSERT(d_morton_keys.size() + 1 == d_deltas.size());
    detail::xor_dispatch(detail::raw(d_morton_keys), d_morton_keys.size(), detail::raw(d_deltas));
}

// Real4 should be float4 or double4.  Throws std::invalid_argument if the number of spheres
// does not exceed d_tree.max_per_leaf (albvh.cuh:795-799).  Resizes d_tree.nodes / leaves to
// the tree that was built (albvh.cuh:842-845).
template <typename Real4, typename DeltaType>
GRACE_HOST void ALBVH_sph(
    const thrust::device_vector<Real4>& d_spheres,
    const thrust::device_vector<DeltaType>& d_deltas,
    Tree& d_tree)
{
    const size_t n = d_spheres.size();
    GRACE_ASSERT(n + 1 == d_deltas.size());
    if (d_tree.leaves.size() < n) d_tree.leaves.resize(n);
    if (d_tree.nodes.size() < 4 * (n - 1)) d_tree.nodes.resize(4 * (n - 1));
    size_t n_leaves = 0;
    detail::albvh_dispatch(detail::raw(d_spheres), n, detail::raw(d_deltas), d_tree.max_per_leaf,
                           reinterpret_cast<int*>(detail::raw(d_tree.nodes)),
                           reinterpret_cast<int*>(detail::raw(d_tree.leaves)),
                           d_tree.root_index_ptr, &n_leaves);
    d_tree.leaves.resize(n_leaves);
    d_tree.nodes.resize(4 * (n_leaves - 1));
}

} // namespace grace
