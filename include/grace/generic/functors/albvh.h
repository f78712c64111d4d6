// grace/generic/functors/albvh.h -- the stock delta functors of the ALBVH build (reference
// include/grace/generic/functors/albvh.h:17-126).  A delta functor is any default-constructible
// type with
//     __host__ __device__ DeltaType operator()(int i, KeyOrPrimitiveIter data, size_t n) const;
// returning the "distance" between elements i and i + 1, and a value no smaller than any real
// delta for i = -1 and i = n - 1 (the two ends are queried by the build).
// grace::compute_deltas(keys_or_prims, deltas, DeltaFunc) accepts these and any of the caller's
// own (grace/cuda/kernels/albvh.cuh).
//
// The arithmetic of the floating-point functors is kept unfused (#pragma clang fp contract(off))
// whatever flags the including translation unit is built with, so that a tree built through the
// generic form equals the one libgrace_hip.so's own delta kernels (csrc/deltas.hip) lead to.
#pragma once

#include "grace/types.h"

#include <iterator>
#include <limits>

namespace grace {

// Morton-key XOR: keys that share a longer prefix are closer (albvh.h:17-49).
struct DeltaXOR
{
    GRACE_HOST_DEVICE uinteger32 operator()(const int i, const uinteger32* morton_keys,
                                            const size_t n_keys) const
    {
        if (i < 0 || size_t(i) + 1 >= n_keys) return uinteger32(-1);
        return morton_keys[i] ^ morton_keys[i + 1];
    }

    GRACE_HOST_DEVICE uinteger64 operator()(const int i, const uinteger64* morton_keys,
                                            const size_t n_keys) const
    {
        if (i < 0 || size_t(i) + 1 >= n_keys) return uinteger64(-1);
        return morton_keys[i] ^ morton_keys[i + 1];
    }
};

// Squared Euclidean distance of neighbouring primitives' .x/.y/.z (albvh.h:51-82: the functor
// holds a CentroidFunc, but the distance it returns is that of the primitives' own leading
// components, formed in their precision and returned as float).
template <typename PrimitiveIter, typename CentroidFunc>
struct DeltaEuclidean
{
    typedef typename std::iterator_traits<PrimitiveIter>::value_type TPrimitive;

    GRACE_HOST_DEVICE DeltaEuclidean() : centroid(CentroidFunc()) {}

    GRACE_HOST_DEVICE float operator()(const int i, PrimitiveIter primitives,
                                       const size_t n_primitives) const
    {
#pragma clang fp contract(off)
        if (i < 0 || size_t(i) + 1 >= n_primitives) return std::numeric_limits<float>::infinity();
        const TPrimitive pi = primitives[i];
        const TPrimitive pj = primitives[i + 1];
        return (pi.x - pj.x) * (pi.x - pj.x) + (pi.y - pj.y) * (pi.y - pj.y)
            + (pi.z - pj.z) * (pi.z - pj.z);
    }

private:
    const CentroidFunc centroid;
};

// Half the surface area of the union of neighbouring primitives' boxes (albvh.h:84-126).
template <typename PrimitiveIter, typename AABBFunc>
struct DeltaSurfaceArea
{
    typedef typename std::iterator_traits<PrimitiveIter>::value_type TPrimitive;

    GRACE_HOST_DEVICE DeltaSurfaceArea() : AABB(AABBFunc()) {}

    GRACE_HOST_DEVICE float operator()(const int i, PrimitiveIter primitives,
                                       const size_t n_primitives) const
    {
#pragma clang fp contract(off)
        if (i < 0 || size_t(i) + 1 >= n_primitives) return std::numeric_limits<float>::infinity();
        float3 boti, topi, botj, topj;
        AABB(primitives[i], &boti, &topi);
        AABB(primitives[i + 1], &botj, &topj);
        const float L_x = fmaxf(topi.x, topj.x) - fminf(boti.x, botj.x);
        const float L_y = fmaxf(topi.y, topj.y) - fminf(boti.y, botj.y);
        const float L_z = fmaxf(topi.z, topj.z) - fminf(boti.z, botj.z);
        return (L_x * L_y) + (L_x * L_z) + (L_y * L_z);
    }

private:
    const AABBFunc AABB;
};

} // namespace grace
