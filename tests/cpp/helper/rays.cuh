// helper/rays.cuh -- the orthogonal -z ray grid of the reference's tests
// (tests/helper/rays.cuh:12-79), composed from grace::orthographic_projection_rays.
#pragma once

#include "grace/cuda/gen_rays.cuh"
#include "grace/ray.h"

#include <thrust/device_vector.h>

inline float3 box_center(const float4 mins, const float4 maxs)
{
    return make_float3((mins.x + maxs.x) / 2., (mins.y + maxs.y) / 2., (mins.z + maxs.z) / 2.);
}

// maxs.w == padding beyond the bounds (e.g. the maximum SPH radius) on all sides.
inline float3 box_span(const float4 mins, const float4 maxs)
{
    return make_float3(maxs.x - mins.x + 2 * maxs.w, maxs.y - mins.y + 2 * maxs.w,
                       maxs.z - mins.z + 2 * maxs.w);
}

inline float per_ray_area(const float3 span, const size_t N_side)
{
    const float cell_x = span.x / N_side;
    const float cell_y = span.y / N_side;
    return cell_x * cell_y;
}

// Rays in the -z direction from the plane z = span.z above the box centre, one per cell of an
// N_side x N_side grid (x fastest), square aspect ratio, length 2 span.z.
inline void orthogonal_rays_z(const size_t N_side, const float4 mins, const float4 maxs,
                              thrust::device_vector<grace::Ray>& d_rays, float* area = NULL)
{
    float3 center = box_center(mins, maxs);
    float3 span = box_span(mins, maxs);
    if (span.x > span.y) { span.y = span.x; }
    else if (span.y > span.x) { span.x = span.y; }
    if (area != NULL) *area = per_ray_area(span, N_side);
    float3 camera_position = make_float3(center.x, center.y, span.z);
    float3 look_at = center;
    float3 view_up = make_float3(0.f, 1.f, 0.f);
    float length = 2 * span.z;
    grace::orthographic_projection_rays(d_rays, N_side, N_side, camera_position, look_at, view_up,
                                        span.y, length);
}
