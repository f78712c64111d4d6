// grace/generic/functors/centroid.h -- centroid functors (reference
// include/grace/generic/functors/centroid.h:14-40).  A centroid functor is any
// default-constructible type with
//     __host__ __device__ float3 operator()(TPrimitive) const;
// grace::morton_keys(prims, ..., CentroidFunc) accepts the stock ones and any of the caller's own
// (grace/cuda/kernels/morton.cuh).
#pragma once

#include "grace/generic/functors/aabb.h"
#include "grace/types.h"

namespace grace {

// The centroid of a primitive = the centroid of its AABB (centroid.h:14-31).
template <typename TPrimitive, typename AABBFunc>
struct PrimitiveCentroid
{
    typedef TPrimitive argument_type;
    typedef float3 result_type;

    GRACE_HOST_DEVICE PrimitiveCentroid() : AABB(AABBFunc()) {}

    GRACE_HOST_DEVICE float3 operator()(TPrimitive primitive) const
    {
        float3 bot, top;
        AABB(primitive, &bot, &top);
        return detail::AABB_centroid(bot, top);
    }

private:
    const AABBFunc AABB;
};

// Spheres {x, y, z, radius}: the centre, narrowed to float (centroid.h:33-40).
struct CentroidSphere
{
    template <typename Real4>
    GRACE_HOST_DEVICE float3 operator()(Real4 sphere) const
    {
        return make_float3(sphere.x, sphere.y, sphere.z);
    }
};

} // namespace grace
