// helper/rays.cuh -- the -z ray grid the test programs project through: N_side x N_side rays
// (x fastest) over the square that contains [mins, maxs] padded by maxs.w on every side, started
// above the box and long enough to cross it (what the reference's tests call orthogonal_rays_z).
// The grid is the library's generator (grace_rays_orthogonal_z, csrc/rays.hip); *area receives
// the area of one ray's cell.
#pragma once

#include "grace/error.h"
#include "grace/ray.h"

#include <thrust/device_vector.h>

inline void orthogonal_rays_z(const size_t N_side, const float4 mins, const float4 maxs,
                              thrust::device_vector<grace::Ray>& d_rays, float* area = NULL)
{
    const float lo[4] = { mins.x, mins.y, mins.z, mins.w }, hi[4] = { maxs.x, maxs.y, maxs.z, maxs.w };
    d_rays.resize(N_side * N_side);
    GRACE_STATUS_CHECK(grace_rays_orthogonal_z(int(N_side), lo, hi, thrust::raw_pointer_cast(d_rays.data()),
                                               area, NULL));
}
