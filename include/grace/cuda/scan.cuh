// grace/cuda/scan.cuh -- per-ray exclusive prefix sums over hit lists (reference
// include/grace/cuda/scan.cuh:15-58, there through the vendored sgpu SegScanCsr; here the
// wave64 segmented scan of libgrace_hip.so, csrc/scan.hip).  Segment s covers
// [offsets[s], offsets[s + 1]) (the last one up to the end of the data); empty segments are
// allowed.
#pragma once

#include "grace/detail/raw.h"

namespace grace {

namespace detail {
inline void segscan_dispatch(const int* off, size_t ns, const float* d, size_t n, float* r)
{ GRACE_STATUS_CHECK(grace_segscan_exclusive_f32(off, ns, d, n, r, NULL)); }
inline void segscan_dispatch(const int* off, size_t ns, const double* d, size_t n, double* r)
{ GRACE_STATUS_CHECK(grace_segscan_exclusive_f64(off, ns, d, n, r, NULL)); }
inline void weights_dispatch(const float* x, size_t n, const float* w, const unsigned int* m, float* out)
{ GRACE_STATUS_CHECK(grace_multiply_by_weights_f32(x, n, w, m, out, NULL)); }
inline void weights_dispatch(const double* x, size_t n, const double* w, const unsigned int* m, double* out)
{ GRACE_STATUS_CHECK(grace_multiply_by_weights_f64(x, n, w, m, out, NULL)); }
} // namespace detail

// d_data and d_results may be the same vector.
template <typename Real>
GRACE_HOST void exclusive_segmented_scan(
    const thrust::device_vector<int>& d_segment_offsets,
    thrust::device_vector<Real>& d_data,
    thrust::device_vector<Real>& d_results)
{
    detail::segscan_dispatch(detail::raw(d_segment_offsets), d_segment_offsets.size(),
                             detail::raw(d_data), d_data.size(), detail::raw(d_results));
}

// weighted_values[i] = d_to_sum[i] * d_weights[d_weight_map[i]], then the exclusive segmented
// sum of the weighted values (scan.cuh:39-58; kernels/weights.cuh:13-27).
// Real: float or double.
template <typename Real>
GRACE_HOST void weighted_exclusive_segmented_scan(
    const thrust::device_vector<Real>& d_to_sum,
    const thrust::device_vector<Real>& d_weights,
    const thrust::device_vector<unsigned int>& d_weight_map,
    const thrust::device_vector<int>& d_segment_offsets,
    thrust::device_vector<Real>& d_sum)
{
    thrust::device_vector<Real> d_weighted(d_to_sum.size());
    detail::weights_dispatch(detail::raw(d_to_sum), d_to_sum.size(), detail::raw(d_weights),
                             detail::raw(d_weight_map), detail::raw(d_weighted));
    grace::exclusive_segmented_scan(d_segment_offsets, d_weighted, d_sum);
}

} // namespace grace
