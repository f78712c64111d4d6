// grace/cuda/sort.cuh -- per-ray sort of hits by distance (reference
// include/grace/cuda/sort.cuh:97-131, there sgpu SegSortPairsFromIndices + thrust::gather; here
// the one-wavefront-per-ray radix sort of libgrace_hip.so, csrc/segsort.hip): within each ray's
// segment the distances become non-decreasing, equal distances keep their order, and the hit
// indices and hit data are permuted by the same map.
#pragma once

#include "grace/detail/raw.h"

namespace grace {

namespace detail {
inline void segsort_dispatch(float* d, const int* off, size_t nr, size_t nh, int* idx, float* data)
{ GRACE_STATUS_CHECK(grace_sort_by_distance_f32(d, off, nr, nh, idx, data, NULL)); }
inline void segsort_dispatch(double* d, const int* off, size_t nr, size_t nh, int* idx, double* data)
{ GRACE_STATUS_CHECK(grace_sort_by_distance_f64(d, off, nr, nh, idx, data, NULL)); }
} // namespace detail

// Real is float or double; IndexType a 32-bit integer; T (the hit data) a type of the width of
// Real (the integrals of trace_sph).
template <typename Real, typename IndexType, typename T>
GRACE_HOST void sort_by_distance(
    thrust::device_vector<Real>& d_hit_distances,
    const thrust::device_vector<int>& d_ray_offsets,
    thrust::device_vector<IndexType>& d_hit_indices,
    thrust::device_vector<T>& d_hit_data)
{
    static_assert(sizeof(IndexType) == sizeof(int), "IndexType must be a 32-bit integer");
    static_assert(sizeof(T) == sizeof(Real), "the hit data must have the width of the distances");
    detail::segsort_dispatch(detail::raw(d_hit_distances), detail::raw(d_ray_offsets),
                             d_ray_offsets.size(), d_hit_distances.size(),
                             reinterpret_cast<int*>(detail::raw(d_hit_indices)),
                             reinterpret_cast<Real*>(detail::raw(d_hit_data)));
}

} // namespace grace
