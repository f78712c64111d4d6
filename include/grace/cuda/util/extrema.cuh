// grace/cuda/util/extrema.cuh -- minima / maxima of arrays of small vectors: the reference's
// grace::min_max_x / _y / _z / _w, min_vec2/3/4 and max_vec2/3/4 (include/grace/cuda/util/
// extrema.cuh:190-772) with their five argument forms each -- device iterator, const / non-const
// device pointer, thrust::device_vector, thrust::host_vector -- and copy_xy / copy_xyz / copy_xyzw
// (extrema.cuh:149-169).  Vec is any type whose leading members x, y (z, w) are contiguous floats,
// doubles, ints or unsigned ints (float2/3/4, double2/3/4, int4 ... or the caller's own struct).
//
// The reference runs one thrust::minmax_element or thrust::reduce per call; here device data goes
// through one streaming pass of libgrace_hip.so (grace_minmax_components, csrc/extrema.hip;
// float4 x y z w together through the fused grace_minmax_f4), host vectors through a plain loop.
// Device iterators must be raw pointers or Thrust device iterators over contiguous storage.
#pragma once

#include "grace/detail/raw.h"

#include <thrust/host_vector.h>

#include <cstddef>
#include <iterator>
#include <type_traits>

namespace grace {

//-----------------------------------------------------------------------------
// Copies between compatible, not necessarily identical, vector types -- e.g. the .x and .y of an
// int3 into the .x and .y of a float4 (extrema.cuh:149-169).
//-----------------------------------------------------------------------------

template <typename Vec2, typename OutType>
GRACE_HOST_DEVICE void copy_xy(const Vec2 src, OutType* const dst)
{
    dst->x = src.x;
    dst->y = src.y;
}

template <typename Vec3, typename OutType>
GRACE_HOST_DEVICE void copy_xyz(const Vec3 src, OutType* const dst)
{
    copy_xy(src, dst);
    dst->z = src.z;
}

template <typename Vec4, typename OutType>
GRACE_HOST_DEVICE void copy_xyzw(const Vec4 src, OutType* const dst)
{
    copy_xyz(src, dst);
    dst->w = src.w;
}

namespace detail {

template <typename E> struct elem_code {
    static_assert(always_false<E>::value,
                  "grace extrema: vector components must be float, double, int or unsigned int");
};
template <> struct elem_code<float> { static const int value = GRACE_ELEM_F32; };
template <> struct elem_code<double> { static const int value = GRACE_ELEM_F64; };
template <> struct elem_code<int> { static const int value = GRACE_ELEM_I32; };
template <> struct elem_code<unsigned int> { static const int value = GRACE_ELEM_U32; };

template <typename Vec>
struct vec_elem { typedef typename std::decay<decltype(std::declval<Vec>().x)>::type type; };

// Minima and maxima of components FIRST .. FIRST + NC - 1 of n device records.
template <int FIRST, int NC, typename Vec>
inline void device_extrema(const Vec* d_data, const size_t N, typename vec_elem<Vec>::type* lo,
                           typename vec_elem<Vec>::type* hi)
{
    typedef typename vec_elem<Vec>::type E;
    static_assert(sizeof(Vec) >= (FIRST + NC) * sizeof(E), "vector type has too few components");
    if constexpr (std::is_same<Vec, float4>::value && FIRST == 0 && NC == 4) {
        GRACE_STATUS_CHECK(grace_minmax_f4(reinterpret_cast<const float*>(d_data), N, lo, hi, NULL));
    } else
    GRACE_STATUS_CHECK(grace_minmax_components(reinterpret_cast<const char*>(d_data) + FIRST * sizeof(E), N,
                                               elem_code<E>::value, NC, sizeof(Vec), lo, hi, NULL));
}

template <int FIRST, int NC, typename VecIter>
inline void host_extrema(VecIter it, const size_t N,
                         typename vec_elem<typename std::iterator_traits<VecIter>::value_type>::type* lo,
                         typename vec_elem<typename std::iterator_traits<VecIter>::value_type>::type* hi)
{
    typedef typename std::iterator_traits<VecIter>::value_type Vec;
    typedef typename vec_elem<Vec>::type E;
    for (size_t i = 0; i < N; ++i, ++it) {
        const Vec v = *it;
        const E* c = reinterpret_cast<const E*>(&v) + FIRST;
        for (int k = 0; k < NC; ++k) {
            if (i == 0 || c[k] < lo[k]) lo[k] = c[k];
            if (i == 0 || hi[k] < c[k]) hi[k] = c[k];
        }
    }
}

// (members .z / .w are only touched for output types that have them)
template <typename OutType, typename E>
inline auto store_z(const E v, OutType* out, int) -> decltype((void)(out->z = v)) { out->z = v; }
template <typename OutType, typename E>
inline void store_z(const E, OutType*, long) {}
template <typename OutType, typename E>
inline auto store_w(const E v, OutType* out, int) -> decltype((void)(out->w = v)) { out->w = v; }
template <typename OutType, typename E>
inline void store_w(const E, OutType*, long) {}

template <int NC, typename OutType, typename E>
inline void store_components(const E (&v)[NC], OutType* out)
{
    out->x = v[0];
    out->y = v[1];
    if constexpr (NC > 2) store_z(v[2], out, 0);
    if constexpr (NC > 3) store_w(v[3], out, 0);
}

} // namespace detail

// ---- one component: min_max_x, min_max_y, min_max_z, min_max_w (extrema.cuh:207-444) --------
// Forms: (device iterator, N), (const Vec* d_data, N), (Vec* d_data, N), device_vector,
// host_vector; *min and *max receive the extrema converted to T.
#define GRACE_MIN_MAX_COMPONENT(NAME, INDEX)                                                      \
    template <typename VecIter, typename T>                                                       \
    GRACE_HOST void NAME(VecIter data_iter, const size_t N, T* min_c, T* max_c)                   \
    {                                                                                             \
        typedef typename std::remove_cv<typename std::iterator_traits<VecIter>::value_type>::type Vec; \
        typename detail::vec_elem<Vec>::type lo, hi;                                              \
        detail::device_extrema<INDEX, 1>(static_cast<const Vec*>(detail::raw_of(data_iter)), N, &lo, &hi); \
        *min_c = lo;                                                                              \
        *max_c = hi;                                                                              \
    }                                                                                             \
    template <typename Vec, typename T>                                                           \
    GRACE_HOST void NAME(const Vec* d_data, const size_t N, T* min_c, T* max_c)                   \
    {                                                                                             \
        typename detail::vec_elem<Vec>::type lo, hi;                                              \
        detail::device_extrema<INDEX, 1>(d_data, N, &lo, &hi);                                    \
        *min_c = lo;                                                                              \
        *max_c = hi;                                                                              \
    }                                                                                             \
    template <typename Vec, typename T>                                                           \
    GRACE_HOST void NAME(Vec* d_data, const size_t N, T* min_c, T* max_c)                         \
    {                                                                                             \
        NAME(static_cast<const Vec*>(d_data), N, min_c, max_c);                                   \
    }                                                                                             \
    template <typename Vec, typename T>                                                           \
    GRACE_HOST void NAME(const thrust::device_vector<Vec>& d_data, T* min_c, T* max_c)            \
    {                                                                                             \
        NAME(detail::raw(d_data), d_data.size(), min_c, max_c);                                   \
    }                                                                                             \
    template <typename Vec, typename T>                                                           \
    GRACE_HOST void NAME(const thrust::host_vector<Vec>& h_data, T* min_c, T* max_c)              \
    {                                                                                             \
        typename detail::vec_elem<Vec>::type lo = 0, hi = 0;                                      \
        detail::host_extrema<INDEX, 1>(h_data.begin(), h_data.size(), &lo, &hi);                  \
        *min_c = lo;                                                                              \
        *max_c = hi;                                                                              \
    }

GRACE_MIN_MAX_COMPONENT(min_max_x, 0)
GRACE_MIN_MAX_COMPONENT(min_max_y, 1)
GRACE_MIN_MAX_COMPONENT(min_max_z, 2)
GRACE_MIN_MAX_COMPONENT(min_max_w, 3)
#undef GRACE_MIN_MAX_COMPONENT

// ---- the leading 2 / 3 / 4 components together: min_vec2/3/4, max_vec2/3/4
//      (extrema.cuh:447-772).  *out receives .x, .y (, .z (, .w)); other members are untouched.
#define GRACE_VEC_EXTREMUM(NAME, NC, WHICH)                                                       \
    template <typename VecIter, typename OutType>                                                 \
    GRACE_HOST void NAME(VecIter data_iter, const size_t N, OutType* out)                         \
    {                                                                                             \
        typedef typename std::remove_cv<typename std::iterator_traits<VecIter>::value_type>::type Vec; \
        typename detail::vec_elem<Vec>::type lo[NC], hi[NC];                                      \
        detail::device_extrema<0, NC>(static_cast<const Vec*>(detail::raw_of(data_iter)), N, lo, hi); \
        detail::store_components<NC>(WHICH, out);                                                 \
    }                                                                                             \
    template <typename Vec, typename OutType>                                                     \
    GRACE_HOST void NAME(const Vec* d_data, const size_t N, OutType* out)                         \
    {                                                                                             \
        typename detail::vec_elem<Vec>::type lo[NC], hi[NC];                                      \
        detail::device_extrema<0, NC>(d_data, N, lo, hi);                                         \
        detail::store_components<NC>(WHICH, out);                                                 \
    }                                                                                             \
    template <typename Vec, typename OutType>                                                     \
    GRACE_HOST void NAME(Vec* d_data, const size_t N, OutType* out)                               \
    {                                                                                             \
        NAME(static_cast<const Vec*>(d_data), N, out);                                            \
    }                                                                                             \
    template <typename Vec, typename OutType>                                                     \
    GRACE_HOST void NAME(const thrust::device_vector<Vec>& d_data, OutType* out)                  \
    {                                                                                             \
        NAME(detail::raw(d_data), d_data.size(), out);                                            \
    }                                                                                             \
    template <typename Vec, typename OutType>                                                     \
    GRACE_HOST void NAME(const thrust::host_vector<Vec>& h_data, OutType* out)                    \
    {                                                                                             \
        typename detail::vec_elem<Vec>::type lo[NC] = {}, hi[NC] = {};                            \
        detail::host_extrema<0, NC>(h_data.begin(), h_data.size(), lo, hi);                       \
        detail::store_components<NC>(WHICH, out);                                                 \
    }

GRACE_VEC_EXTREMUM(min_vec2, 2, lo)
GRACE_VEC_EXTREMUM(min_vec3, 3, lo)
GRACE_VEC_EXTREMUM(min_vec4, 4, lo)
GRACE_VEC_EXTREMUM(max_vec2, 2, hi)
GRACE_VEC_EXTREMUM(max_vec3, 3, hi)
GRACE_VEC_EXTREMUM(max_vec4, 4, hi)
#undef GRACE_VEC_EXTREMUM

namespace detail {
// Both bounds of the x, y, z of n device records in ONE pass (the bounds-free morton_keys forms).
template <typename Vec, typename OutType>
inline void min_max_vec3(const Vec* d_data, const size_t N, OutType* mins, OutType* maxs)
{
    typename vec_elem<Vec>::type lo[3], hi[3];
    device_extrema<0, 3>(d_data, N, lo, hi);
    store_components<3>(lo, mins);
    store_components<3>(hi, maxs);
}
} // namespace detail

} // namespace grace
