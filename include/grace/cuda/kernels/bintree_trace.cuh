// grace/cuda/kernels/bintree_trace.cuh -- grace::trace / grace::trace_texref, the reference's
// generic functor-parameterised traversal (include/grace/cuda/kernels/bintree_trace.cuh:214-367),
// with its signatures over thrust::device_vector.  User functors cannot cross the C ABI of
// libgrace_hip.so, so this one entry point is a header-template HIP kernel
// (grace/hip/trace_core.hpp: the plain packet walk with the reference's functor contract;
// the built-in SPH / triangle instantiations behind the library are the tuned ones).
#pragma once

#include "grace/cuda/nodes.h"
#include "grace/detail/raw.h"
#include "grace/hip/trace_core.hpp"

namespace grace {

// raw-pointer form, bintree_trace.cuh:214-284
template <typename RayData, typename TPrimitive, typename Init, typename Intersection,
          typename OnHit, typename OnRayEntry, typename OnRayExit>
GRACE_HOST void trace(const Ray* d_rays, const size_t N_rays, const TPrimitive* d_prims,
                      const size_t /*N_primitives*/, const Tree& d_tree,
                      const size_t user_smem_bytes, Init init, Intersection intersect, OnHit on_hit,
                      OnRayEntry ray_entry, OnRayExit ray_exit)
{
    detail::trace_launch<RayData>(d_rays, N_rays, d_prims, detail::raw(d_tree.nodes),
                                  d_tree.leaves.size() - 1, detail::raw(d_tree.leaves),
                                  d_tree.root_index_ptr, user_smem_bytes, init, intersect, on_hit,
                                  ray_entry, ray_exit);
}

// device_vector form, bintree_trace.cuh:286-315
template <typename RayData, typename TPrimitive, typename Init, typename Intersection,
          typename OnHit, typename OnRayEntry, typename OnRayExit>
GRACE_HOST void trace(const thrust::device_vector<Ray>& d_rays,
                      const thrust::device_vector<TPrimitive>& d_primitives, const Tree& d_tree,
                      const size_t user_smem_bytes, Init init, Intersection intersect, OnHit on_hit,
                      OnRayEntry ray_entry, OnRayExit ray_exit)
{
    trace<RayData>(detail::raw(d_rays), d_rays.size(), detail::raw(d_primitives),
                   d_primitives.size(), d_tree, user_smem_bytes, init, intersect, on_hit, ray_entry,
                   ray_exit);
}

// bintree_trace.cuh:317-367: the same signature (the reference binds the primitives to a texture
// reference first; there are none on this side).
template <typename RayData, typename TPrimitive, typename Init, typename Intersection,
          typename OnHit, typename OnRayEntry, typename OnRayExit>
GRACE_HOST void trace_texref(const thrust::device_vector<Ray>& d_rays,
                             const thrust::device_vector<TPrimitive>& d_primitives,
                             const Tree& d_tree, const size_t user_smem_bytes, Init init,
                             Intersection intersect, OnHit on_hit, OnRayEntry ray_entry,
                             OnRayExit ray_exit)
{
    trace<RayData>(d_rays, d_primitives, d_tree, user_smem_bytes, init, intersect, on_hit,
                   ray_entry, ray_exit);
}

} // namespace grace
