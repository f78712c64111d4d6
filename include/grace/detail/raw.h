// grace/detail/raw.h -- helpers shared by the drop-in headers: raw device pointers of
// thrust vectors, and compile-time tags for the (Real4, key, delta) types the C ABI covers.
#pragma once

#include "grace/error.h"
#include "grace/types.h"

#include <thrust/device_vector.h>

namespace grace {
namespace detail {

template <typename T>
inline T* raw(thrust::device_vector<T>& v) { return thrust::raw_pointer_cast(v.data()); }
template <typename T>
inline const T* raw(const thrust::device_vector<T>& v) { return thrust::raw_pointer_cast(v.data()); }

// Raw device pointer behind an iterator: a plain pointer, or a Thrust device iterator /
// device_ptr over contiguous storage.
template <typename T>
inline T* raw_of(T* p) { return p; }
template <typename Iter>
inline auto raw_of(Iter it) -> decltype(thrust::raw_pointer_cast(&*it)) { return thrust::raw_pointer_cast(&*it); }

template <typename T> struct always_false { static const bool value = false; };

template <typename Real4> struct is_float4 { static const bool value = false; };
template <> struct is_float4<float4> { static const bool value = true; };
template <typename Real4> struct is_double4 { static const bool value = false; };
template <> struct is_double4<double4> { static const bool value = true; };

// x y z of float3 / double3 / float4 / ... as an array of the component type.
template <typename Real, typename Vec3>
inline void xyz(const Vec3& v, Real* out) { out[0] = Real(v.x); out[1] = Real(v.y); out[2] = Real(v.z); }

} // namespace detail
} // namespace grace
