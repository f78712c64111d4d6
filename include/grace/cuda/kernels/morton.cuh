// grace/cuda/kernels/morton.cuh -- the generic grace::morton_keys forms of the reference
// (include/grace/cuda/kernels/morton.cuh:97-189): Morton keys of any primitive type through a
// caller-supplied centroid functor, with the bounds given or computed.
//
// A functor cannot cross the C ABI of libgrace_hip.so, so the functor-dependent part -- one
// centroid per primitive -- runs in a header kernel compiled into the caller's translation unit
// (the role compute_centroids_kernel plays in the reference, kernels/aabb.cuh:14-48); bounds and
// keys are then the library's kernels over the float3 centroids (csrc/morton.hip: the same
// scale-and-truncate arithmetic in the precision of Real3, morton.cuh:43-50,104-113).  Spheres
// with the stock CentroidSphere skip the intermediate array and go straight to the library's
// fused kernels.  KeyIter's value type selects 30-bit (<= 32 bits wide) or 63-bit keys
// (morton.cuh:106-108).  Iterators must be raw device pointers or Thrust device iterators
// over contiguous storage.
#pragma once

#include "grace/cuda/util/extrema.cuh"
#include "grace/detail/raw.h"
#include "grace/error.h"
#include "grace/generic/bits.h"
#include "grace/generic/functors/aabb.h"
#include "grace/generic/functors/centroid.h"
#include "grace/generic/morton.h"
#include "grace/types.h"

#include <thrust/device_vector.h>

#include <climits>
#include <iterator>
#include <type_traits>

namespace grace {

namespace detail {

// One centroid per primitive, by the caller's functor.
template <typename PrimitiveIter, typename CentroidFunc>
__global__ __launch_bounds__(256) void centroids_kernel(PrimitiveIter primitives, const size_t n,
                                                        float3* __restrict__ centroids,
                                                        const CentroidFunc centroid)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n;
         i += size_t(gridDim.x) * blockDim.x)
        centroids[i] = centroid(primitives[i]);
}

template <typename PrimitiveIter, typename CentroidFunc>
inline void compute_centroids(PrimitiveIter d_prims_iter, const size_t n, float3* d_centroids,
                              const CentroidFunc centroid)
{
    if (n == 0) return;
    const size_t blocks = (n + 255) / 256;
    centroids_kernel<<<unsigned(blocks < 4096 ? blocks : 4096), 256>>>(d_prims_iter, n, d_centroids,
                                                                       centroid);
    GRACE_HIP_CHECK(hipGetLastError());
}

// (bounds precision, key width) -> the library's key kernel over n float3 points
inline void point_keys(const float3* c, size_t n, const float* b, const float* t, uinteger32* k)
{ GRACE_STATUS_CHECK(grace_morton_keys30_points(c, n, 0, 3, b, t, k, NULL)); }
inline void point_keys(const float3* c, size_t n, const float* b, const float* t, uinteger64* k)
{ GRACE_STATUS_CHECK(grace_morton_keys63_points(c, n, 0, 3, b, t, k, NULL)); }
inline void point_keys(const float3* c, size_t n, const double* b, const double* t, uinteger32* k)
{ GRACE_STATUS_CHECK(grace_morton_keys30_points_d3(c, n, 0, 3, b, t, k, NULL)); }
inline void point_keys(const float3* c, size_t n, const double* b, const double* t, uinteger64* k)
{ GRACE_STATUS_CHECK(grace_morton_keys63_points_d3(c, n, 0, 3, b, t, k, NULL)); }

// Spheres with the stock centroid: n records of four floats / doubles, no intermediate array.
inline void sphere_keys(const float4* s, size_t n, const float* b, const float* t, uinteger32* k)
{ GRACE_STATUS_CHECK(grace_morton_keys30_f4(&s->x, n, b, t, k, NULL)); }
inline void sphere_keys(const float4* s, size_t n, const float* b, const float* t, uinteger64* k)
{ GRACE_STATUS_CHECK(grace_morton_keys63_f4(&s->x, n, b, t, k, NULL)); }
inline void sphere_keys(const float4* s, size_t n, const double* b, const double* t, uinteger32* k)
{ GRACE_STATUS_CHECK(grace_morton_keys30_f4_d3(&s->x, n, b, t, k, NULL)); }
inline void sphere_keys(const float4* s, size_t n, const double* b, const double* t, uinteger64* k)
{ GRACE_STATUS_CHECK(grace_morton_keys63_f4_d3(&s->x, n, b, t, k, NULL)); }
inline void sphere_keys(const double4* s, size_t n, const float* b, const float* t, uinteger32* k)
{ GRACE_STATUS_CHECK(grace_morton_keys30_points(s, n, 1, 4, b, t, k, NULL)); }
inline void sphere_keys(const double4* s, size_t n, const float* b, const float* t, uinteger64* k)
{ GRACE_STATUS_CHECK(grace_morton_keys63_points(s, n, 1, 4, b, t, k, NULL)); }
inline void sphere_keys(const double4* s, size_t n, const double* b, const double* t, uinteger32* k)
{ GRACE_STATUS_CHECK(grace_morton_keys30_points_d3(s, n, 1, 4, b, t, k, NULL)); }
inline void sphere_keys(const double4* s, size_t n, const double* b, const double* t, uinteger64* k)
{ GRACE_STATUS_CHECK(grace_morton_keys63_points_d3(s, n, 1, 4, b, t, k, NULL)); }

// The key type the library writes for a caller's KeyType: uinteger32 for types up to 32 bits
// wide, uinteger64 above (morton.cuh:106-108); the caller's type must have that width.
template <typename KeyType>
struct key_word {
    static_assert(std::is_integral<KeyType>::value && (sizeof(KeyType) == 4 || sizeof(KeyType) == 8),
                  "grace::morton_keys: KeyType must be a 32- or 64-bit integer type "
                  "(grace::uinteger32 / grace::uinteger64)");
    typedef typename std::conditional<sizeof(KeyType) == 4, uinteger32, uinteger64>::type type;
};

template <typename T> struct is_stock_sphere
{ static const bool value = std::is_same<T, float4>::value || std::is_same<T, double4>::value; };

// Scalar type of the bounds: Real3 is any type with .x/.y/.z; the key arithmetic runs in the
// precision of its components (morton.cuh:104-113).
template <typename Real3>
struct component_type { typedef typename std::decay<decltype(Real3().x)>::type type; };

} // namespace detail

// Morton keys given the box containing every centroid.
template <typename PrimitiveIter, typename Real3, typename KeyIter, typename CentroidFunc>
GRACE_HOST void morton_keys(
    PrimitiveIter d_prims_iter,
    const size_t N_primitives,
    const Real3 AABB_bot,
    const Real3 AABB_top,
    KeyIter d_keys_iter,
    const CentroidFunc centroid)
{
    typedef typename std::iterator_traits<PrimitiveIter>::value_type TPrimitive;
    typedef typename std::iterator_traits<KeyIter>::value_type KeyType;
    typedef typename detail::key_word<KeyType>::type KeyWord;
    typedef typename detail::component_type<Real3>::type R;
    typedef typename std::conditional<std::is_same<R, double>::value, double, float>::type B;

    B bot[3], top[3];
    detail::xyz(AABB_bot, bot);
    detail::xyz(AABB_top, top);
    KeyWord* keys = reinterpret_cast<KeyWord*>(detail::raw_of(d_keys_iter));

    if constexpr (detail::is_stock_sphere<typename std::remove_cv<TPrimitive>::type>::value
                  && std::is_same<CentroidFunc, CentroidSphere>::value) {
        detail::sphere_keys(detail::raw_of(d_prims_iter), N_primitives, bot, top, keys);
    } else {
        thrust::device_vector<float3> d_centroids(N_primitives);
        detail::compute_centroids(d_prims_iter, N_primitives, detail::raw(d_centroids), centroid);
        detail::point_keys(detail::raw(d_centroids), N_primitives, bot, top, keys);
    }
}

template <typename TPrimitive, typename Real3, typename KeyType, typename CentroidFunc>
GRACE_HOST void morton_keys(
    const thrust::device_vector<TPrimitive>& d_primitives,
    const Real3 AABB_bot,
    const Real3 AABB_top,
    thrust::device_vector<KeyType>& d_keys,
    const CentroidFunc centroid)
{
    morton_keys(detail::raw(d_primitives), d_primitives.size(), AABB_bot, AABB_top,
                detail::raw(d_keys), centroid);
}

// ... additionally computing that box: the bounds of the centroids (float3), optionally
// returned through bots / tops.  O(N_primitives) temporary storage.
template <typename PrimitiveIter, typename KeyIter, typename CentroidFunc>
GRACE_HOST void morton_keys(
    PrimitiveIter d_prims_iter,
    const size_t N_primitives,
    KeyIter d_keys_iter,
    const CentroidFunc centroid,
    float3* const bots = NULL,
    float3* const tops = NULL)
{
    typedef typename std::iterator_traits<KeyIter>::value_type KeyType;
    typedef typename detail::key_word<KeyType>::type KeyWord;

    thrust::device_vector<float3> d_centroids(N_primitives);
    detail::compute_centroids(d_prims_iter, N_primitives, detail::raw(d_centroids), centroid);

    float3 mins, maxs;
    detail::min_max_vec3(detail::raw(d_centroids), N_primitives, &mins, &maxs);   // one pass

    float bot[3], top[3];
    detail::xyz(mins, bot);
    detail::xyz(maxs, top);
    detail::point_keys(detail::raw(d_centroids), N_primitives, bot, top,
                       reinterpret_cast<KeyWord*>(detail::raw_of(d_keys_iter)));

    if (bots != NULL) *bots = mins;
    if (tops != NULL) *tops = maxs;
}

template <typename TPrimitive, typename KeyType, typename CentroidFunc>
GRACE_HOST void morton_keys(
    const thrust::device_vector<TPrimitive>& d_primitives,
    thrust::device_vector<KeyType>& d_keys,
    const CentroidFunc centroid,
    float3* const bots = NULL,
    float3* const tops = NULL)
{
    morton_keys(detail::raw(d_primitives), d_primitives.size(), detail::raw(d_keys), centroid,
                bots, tops);
}

} // namespace grace
