// grace/generic/functors/aabb.h -- the stock AABB functor for spheres and the centroid of a box
// (reference include/grace/generic/functors/aabb.h:9-42).  An AABB functor is any
// default-constructible type with
//     __host__ __device__ void operator()(TPrimitive, float3* bot, float3* top) const;
// grace::build_ALBVH(tree, prims, deltas, AABBFunc) accepts the stock one and any of the caller's
// own (grace/cuda/kernels/albvh.cuh).
#pragma once

#include "grace/types.h"

#include <iterator>

namespace grace {

struct AABBSphere
{
    // sphere = {x, y, z, radius}; float4 or double4.  The corners are narrowed to float, as the
    // tree stores them (aabb.h:12-25).
    template <typename Real4>
    GRACE_HOST_DEVICE void operator()(Real4 sphere, float3* bot, float3* top) const
    {
        bot->x = sphere.x - sphere.w; top->x = sphere.x + sphere.w;
        bot->y = sphere.y - sphere.w; top->y = sphere.y + sphere.w;
        bot->z = sphere.z - sphere.w; top->z = sphere.z + sphere.w;
    }
};

namespace detail {

// Mid-point of a box, formed in double and narrowed (aabb.h:31-39).
GRACE_HOST_DEVICE float3 AABB_centroid(const float3 bot, const float3 top)
{
    float3 centre;
    centre.x = (static_cast<double>(bot.x) + top.x) / 2.;
    centre.y = (static_cast<double>(bot.y) + top.y) / 2.;
    centre.z = (static_cast<double>(bot.z) + top.z) / 2.;
    return centre;
}

} // namespace detail

} // namespace grace
