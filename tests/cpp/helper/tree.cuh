// helper/tree.cuh -- build_tree / random_spheres_tree for the test programs: sort the spheres
// along the 30-bit Morton curve (inside the given box, or inside the box of their centres), take
// the Euclidean deltas of neighbours, build the ALBVH -- the composition the reference's tests
// use (tests/helper/tree.cuh), written against the drop-in API.
#pragma once

#include "grace/cuda/build_sph.cuh"
#include "grace/cuda/nodes.h"
#include "grace/generic/meta.h"

#include "helper/random.cuh"

#include <thrust/device_vector.h>

namespace helper_detail {
template <typename Real4>
inline void deltas_and_tree(const thrust::device_vector<Real4>& sorted_spheres, grace::Tree& tree)
{
    thrust::device_vector<typename grace::Real4ToRealMapper<Real4>::type> deltas(sorted_spheres.size() + 1);
    grace::euclidean_deltas_sph(sorted_spheres, deltas);
    grace::ALBVH_sph(sorted_spheres, deltas, tree);
}
} // namespace helper_detail

template <typename Real4>
void build_tree(thrust::device_vector<Real4>& spheres, grace::Tree& tree)
{
    grace::morton_keys30_sort_sph(spheres);
    helper_detail::deltas_and_tree(spheres, tree);
}

// low / high: anything with .x, .y, .z (the key arithmetic runs in float).
template <typename Real3, typename Real4>
void build_tree(thrust::device_vector<Real4>& spheres, const Real3 low, const Real3 high,
                grace::Tree& tree)
{
    grace::morton_keys30_sort_sph(spheres, make_float3(low.x, low.y, low.z),
                                  make_float3(high.x, high.y, high.z));
    helper_detail::deltas_and_tree(spheres, tree);
}

template <typename Real4>
void random_spheres_tree(const Real4 low, const Real4 high, const size_t N,
                         thrust::device_vector<Real4>& spheres, grace::Tree& tree)
{
    random_real4(low, high, N, spheres);
    build_tree(spheres, low, high, tree);
}
