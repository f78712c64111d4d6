// grace/hip/trace.hpp -- grace::trace / grace::trace_texref (reference
// include/grace/cuda/kernels/bintree_trace.cuh:214-367) over the containers of the HIP-free
// mirror include/grace/grace.h; kernel, functor contract and stock functors are in
// grace/hip/trace_core.hpp.  Compile the including file with hipcc --offload-arch=gfx950.
#pragma once

#include "grace/grace.h"
#include "grace/hip/trace_core.hpp"

namespace grace {

// raw-pointer form, bintree_trace.cuh:214-284
template <typename RayData, typename TPrimitive, typename Init, typename Intersection,
          typename OnHit, typename OnRayEntry, typename OnRayExit>
inline void trace(const Ray* d_rays, const size_t N_rays, const TPrimitive* d_prims,
                  const size_t /*N_primitives*/, const Tree& d_tree, const size_t user_smem_bytes,
                  Init init, Intersection intersect, OnHit on_hit, OnRayEntry ray_entry,
                  OnRayExit ray_exit)
{
    detail::trace_launch<RayData>(d_rays, N_rays, d_prims, d_tree.nodes.data(),
                                  d_tree.leaves.size() - 1, d_tree.leaves.data(),
                                  d_tree.root_index_ptr, user_smem_bytes, init, intersect, on_hit,
                                  ray_entry, ray_exit);
}

// device_vector form, bintree_trace.cuh:286-315 (trace_texref has the same signature; there
// are no texture references on this side).
template <typename RayData, typename TPrimitive, typename Init, typename Intersection,
          typename OnHit, typename OnRayEntry, typename OnRayExit>
inline void trace(const device_vector<Ray>& d_rays, const device_vector<TPrimitive>& d_primitives,
                  const Tree& d_tree, const size_t user_smem_bytes, Init init,
                  Intersection intersect, OnHit on_hit, OnRayEntry ray_entry, OnRayExit ray_exit)
{
    trace<RayData>(d_rays.data(), d_rays.size(), d_primitives.data(), d_primitives.size(), d_tree,
                   user_smem_bytes, init, intersect, on_hit, ray_entry, ray_exit);
}

template <typename RayData, typename TPrimitive, typename Init, typename Intersection,
          typename OnHit, typename OnRayEntry, typename OnRayExit>
inline void trace_texref(const device_vector<Ray>& d_rays,
                         const device_vector<TPrimitive>& d_primitives, const Tree& d_tree,
                         const size_t user_smem_bytes, Init init, Intersection intersect,
                         OnHit on_hit, OnRayEntry ray_entry, OnRayExit ray_exit)
{
    trace<RayData>(d_rays, d_primitives, d_tree, user_smem_bytes, init, intersect, on_hit,
                   ray_entry, ray_exit);
}

} // namespace grace
