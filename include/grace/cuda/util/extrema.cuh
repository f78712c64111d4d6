// grace/cuda/util/extrema.cuh -- component-wise minima / maxima of device vectors of float4
// (reference include/grace/cuda/util/extrema.cuh:555-607, 720-772, as called by
// tests/project_gadget/project_gadget.cu:66-68).  The reference runs one thrust::reduce per
// call; here one fused pass of libgrace_hip.so (grace_minmax_f4) serves both.
#pragma once

#include "grace/detail/raw.h"

namespace grace {

namespace detail {
inline void minmax_f4(const float4* d_data, size_t N, float* lo, float* hi)
{
    GRACE_STATUS_CHECK(grace_minmax_f4(reinterpret_cast<const float*>(d_data), N, lo, hi, NULL));
}
} // namespace detail

// d_data must be a pointer to DEVICE memory.
template <typename OutType>
GRACE_HOST void min_vec4(const float4* d_data, const size_t N, OutType* mins)
{
    float lo[4], hi[4];
    detail::minmax_f4(d_data, N, lo, hi);
    mins->x = lo[0]; mins->y = lo[1]; mins->z = lo[2]; mins->w = lo[3];
}

template <typename OutType>
GRACE_HOST void max_vec4(const float4* d_data, const size_t N, OutType* maxs)
{
    float lo[4], hi[4];
    detail::minmax_f4(d_data, N, lo, hi);
    maxs->x = hi[0]; maxs->y = hi[1]; maxs->z = hi[2]; maxs->w = hi[3];
}

template <typename OutType>
GRACE_HOST void min_vec4(const thrust::device_vector<float4>& d_data, OutType* mins)
{
    min_vec4(detail::raw(d_data), d_data.size(), mins);
}

template <typename OutType>
GRACE_HOST void max_vec4(const thrust::device_vector<float4>& d_data, OutType* maxs)
{
    max_vec4(detail::raw(d_data), d_data.size(), maxs);
}

// x, y, z only (extrema.cuh:500-552, 665-717).
template <typename OutType>
GRACE_HOST void min_vec3(const thrust::device_vector<float4>& d_data, OutType* mins)
{
    float lo[4], hi[4];
    detail::minmax_f4(detail::raw(d_data), d_data.size(), lo, hi);
    mins->x = lo[0]; mins->y = lo[1]; mins->z = lo[2];
}

template <typename OutType>
GRACE_HOST void max_vec3(const thrust::device_vector<float4>& d_data, OutType* maxs)
{
    float lo[4], hi[4];
    detail::minmax_f4(detail::raw(d_data), d_data.size(), lo, hi);
    maxs->x = hi[0]; maxs->y = hi[1]; maxs->z = hi[2];
}

} // namespace grace
