// grace/cuda/kernels/albvh.cuh -- the generic build forms of the reference
// (include/grace/cuda/kernels/albvh.cuh:949-1072): grace::compute_deltas with any delta functor and
// grace::build_ALBVH with any AABB functor and DeltaComp = thrust::less (default) or
// thrust::greater, over raw device pointers / Thrust device iterators or device_vectors.
//
// A functor cannot cross the C ABI of libgrace_hip.so, so the functor-dependent part runs in a
// header kernel compiled into the caller's translation unit -- one delta per neighbouring pair
// (the reference's compute_deltas_kernel, albvh.cuh:33-47), one box per primitive -- and the
// tree itself is built by the library from those boxes (grace_albvh_build_ex, csrc/albvh.hip:
// leaf and node boxes are min / max unions of primitive boxes, so the floats are those of the
// reference's bottom-up propagation).  The stock combinations (DeltaXOR over keys; spheres with
// AABBSphere) skip the intermediate arrays and go straight to the library's fused kernels.
// Any other comparator than less / greater is refused at compile time.
#pragma once

#include "grace/cuda/nodes.h"
#include "grace/detail/raw.h"
#include "grace/error.h"
#include "grace/generic/functors/aabb.h"
#include "grace/generic/functors/albvh.h"
#include "grace/types.h"

#include <thrust/device_vector.h>
#include <thrust/functional.h>

#include <functional>
#include <iterator>
#include <type_traits>

namespace grace {

namespace detail {

// deltas[i] = delta_func(i - 1, data, n), i in [0, n]  (albvh.cuh:33-47)
template <typename KeyIter, typename DeltaIter, typename DeltaFunc>
__global__ __launch_bounds__(256) void deltas_kernel(KeyIter keys, const size_t n_keys,
                                                     DeltaIter deltas, const DeltaFunc delta_func)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i <= n_keys;
         i += size_t(gridDim.x) * blockDim.x)
        deltas[i] = delta_func(int(i) - 1, keys, n_keys);
}

// boxes[6 i ..] = {bot.x, bot.y, bot.z, top.x, top.y, top.z} of primitive i, by the caller's functor
template <typename PrimitiveIter, typename AABBFunc>
__global__ __launch_bounds__(256) void prim_boxes_kernel(PrimitiveIter primitives, const size_t n,
                                                         float* __restrict__ boxes,
                                                         const AABBFunc AABB)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n;
         i += size_t(gridDim.x) * blockDim.x) {
        float3 bot, top;
        AABB(primitives[i], &bot, &top);
        float* b = boxes + 6 * i;
        b[0] = bot.x; b[1] = bot.y; b[2] = bot.z;
        b[3] = top.x; b[4] = top.y; b[5] = top.z;
    }
}

inline unsigned grid_for(const size_t n)
{
    const size_t blocks = (n + 255) / 256;
    return unsigned(blocks < 4096 ? (blocks ? blocks : 1) : 4096);
}

template <typename DeltaType> struct delta_code {
    static_assert(always_false<DeltaType>::value,
                  "grace::build_ALBVH: deltas must be float, double, grace::uinteger32 or grace::uinteger64");
};
template <> struct delta_code<float> { static const int value = GRACE_DELTA_F32; };
template <> struct delta_code<double> { static const int value = GRACE_DELTA_F64; };
template <> struct delta_code<uinteger32> { static const int value = GRACE_DELTA_U32; };
template <> struct delta_code<uinteger64> { static const int value = GRACE_DELTA_U64; };

template <typename DeltaComp, typename DeltaType> struct comp_code {
    static_assert(always_false<DeltaComp>::value,
                  "grace::build_ALBVH: DeltaComp must be thrust::less<DeltaType> (the default) or "
                  "thrust::greater<DeltaType> (std::less / std::greater are accepted too): the tree "
                  "builder lives behind the C ABI of libgrace_hip.so, which a caller-defined "
                  "comparator cannot cross");
};
template <typename D> struct comp_code<thrust::less<D>, D> { static const int value = GRACE_COMP_LESS; };
template <typename D> struct comp_code<thrust::greater<D>, D> { static const int value = GRACE_COMP_GREATER; };
template <typename D> struct comp_code<std::less<D>, D> { static const int value = GRACE_COMP_LESS; };
template <typename D> struct comp_code<std::greater<D>, D> { static const int value = GRACE_COMP_GREATER; };

template <typename T> struct stock_prim_code { static const int value = -1; };
template <> struct stock_prim_code<float4> { static const int value = GRACE_PRIM_SPHERE_F4; };
template <> struct stock_prim_code<double4> { static const int value = GRACE_PRIM_SPHERE_D4; };

} // namespace detail

//-----------------------------------------------------------------------------
// User functions for ALBVH building.
//-----------------------------------------------------------------------------

template <typename KeyIter, typename DeltaIter, typename DeltaFunc>
GRACE_HOST void compute_deltas(
    KeyIter d_keys_iter,
    const size_t N_keys,
    DeltaIter d_deltas_iter,
    const DeltaFunc delta_func)
{
    typedef typename std::remove_cv<typename std::iterator_traits<KeyIter>::value_type>::type KeyType;
    typedef typename std::iterator_traits<DeltaIter>::value_type DeltaType;
    if constexpr (std::is_same<DeltaFunc, DeltaXOR>::value && std::is_same<KeyType, uinteger32>::value
                  && std::is_same<DeltaType, uinteger32>::value) {
        GRACE_STATUS_CHECK(grace_deltas_xor_u32(detail::raw_of(d_keys_iter), N_keys,
                                                detail::raw_of(d_deltas_iter), NULL));
    } else if constexpr (std::is_same<DeltaFunc, DeltaXOR>::value && std::is_same<KeyType, uinteger64>::value
                         && std::is_same<DeltaType, uinteger64>::value) {
        GRACE_STATUS_CHECK(grace_deltas_xor_u64(detail::raw_of(d_keys_iter), N_keys,
                                                detail::raw_of(d_deltas_iter), NULL));
    } else {
        detail::deltas_kernel<<<detail::grid_for(N_keys + 1), 256>>>(d_keys_iter, N_keys,
                                                                      d_deltas_iter, delta_func);
        GRACE_HIP_CHECK(hipGetLastError());
    }
}

template <typename KeyType, typename DeltaType, typename DeltaFunc>
GRACE_HOST void compute_deltas(
    const thrust::device_vector<KeyType>& d_keys,
    thrust::device_vector<DeltaType>& d_deltas,
    const DeltaFunc delta_func)
{
    GRACE_ASSERT(d_keys.size() + 1 == d_deltas.size());
    compute_deltas(detail::raw(d_keys), d_keys.size(), detail::raw(d_deltas), delta_func);
}

// The number of primitives is d_tree.leaves.size() (a Tree is sized for one leaf per primitive,
// nodes.h; albvh.cuh:1005).  Throws std::invalid_argument if it does not exceed
// d_tree.max_per_leaf (albvh.cuh:795-799); shrinks d_tree.nodes / leaves to the tree that was
// built (albvh.cuh:842-845).
template <typename PrimitiveIter, typename DeltaIter, typename DeltaComp, typename AABBFunc>
GRACE_HOST void build_ALBVH(
    Tree& d_tree,
    PrimitiveIter d_prims_iter,
    DeltaIter d_deltas_iter,
    const DeltaComp /*delta_comp*/,
    const AABBFunc AABB,
    const bool wipe = false)
{
    typedef typename std::remove_cv<typename std::iterator_traits<PrimitiveIter>::value_type>::type TPrimitive;
    typedef typename std::remove_cv<typename std::iterator_traits<DeltaIter>::value_type>::type DeltaType;
    static_assert(sizeof(int4) == sizeof(float4), "node records are four 16-byte words");

    const size_t n = d_tree.leaves.size();
    if (d_tree.nodes.size() < 4 * (n - 1)) d_tree.nodes.resize(4 * (n - 1));
    if (wipe) {
        GRACE_STATUS_CHECK(grace_memset(detail::raw(d_tree.nodes), 0, d_tree.nodes.size() * sizeof(int4), NULL));
        GRACE_STATUS_CHECK(grace_memset(detail::raw(d_tree.leaves), 0, d_tree.leaves.size() * sizeof(int4), NULL));
    }

    const int delta_type = detail::delta_code<DeltaType>::value;
    const int comp = detail::comp_code<DeltaComp, DeltaType>::value;
    const void* deltas = detail::raw_of(d_deltas_iter);
    int* nodes = reinterpret_cast<int*>(detail::raw(d_tree.nodes));
    int* leaves = reinterpret_cast<int*>(detail::raw(d_tree.leaves));
    size_t n_leaves = 0;

    if constexpr (detail::stock_prim_code<TPrimitive>::value >= 0 && std::is_same<AABBFunc, AABBSphere>::value) {
        GRACE_STATUS_CHECK(grace_albvh_build_ex(detail::stock_prim_code<TPrimitive>::value,
                                                detail::raw_of(d_prims_iter), n, delta_type, deltas, comp,
                                                d_tree.max_per_leaf, nodes, leaves, d_tree.root_index_ptr,
                                                &n_leaves, NULL));
    } else {
        thrust::device_vector<float> d_boxes(6 * n);
        if (n) {
            detail::prim_boxes_kernel<<<detail::grid_for(n), 256>>>(d_prims_iter, n, detail::raw(d_boxes), AABB);
            GRACE_HIP_CHECK(hipGetLastError());
        }
        GRACE_STATUS_CHECK(grace_albvh_build_ex(GRACE_PRIM_BOX, detail::raw(d_boxes), n, delta_type, deltas,
                                                comp, d_tree.max_per_leaf, nodes, leaves,
                                                d_tree.root_index_ptr, &n_leaves, NULL));
    }
    d_tree.leaves.resize(n_leaves);
    d_tree.nodes.resize(4 * (n_leaves - 1));
}

template <typename TPrimitive, typename DeltaType, typename DeltaComp, typename AABBFunc>
GRACE_HOST void build_ALBVH(
    Tree& d_tree,
    const thrust::device_vector<TPrimitive>& d_primitives,
    const thrust::device_vector<DeltaType>& d_deltas,
    const DeltaComp delta_comp,
    const AABBFunc AABB,
    const bool wipe = false)
{
    build_ALBVH(d_tree, detail::raw(d_primitives), detail::raw(d_deltas), delta_comp, AABB, wipe);
}

// Specialized with DeltaComp = thrust::less<DeltaType>
template <typename PrimitiveIter, typename DeltaIter, typename AABBFunc>
GRACE_HOST void build_ALBVH(
    Tree& d_tree,
    PrimitiveIter d_prims_iter,
    DeltaIter d_deltas_iter,
    const AABBFunc AABB,
    const bool wipe = false)
{
    typedef typename std::remove_cv<typename std::iterator_traits<DeltaIter>::value_type>::type DeltaType;
    build_ALBVH(d_tree, d_prims_iter, d_deltas_iter, thrust::less<DeltaType>(), AABB, wipe);
}

// Specialized with DeltaComp = thrust::less<DeltaType>
template <typename TPrimitive, typename DeltaType, typename AABBFunc>
GRACE_HOST void build_ALBVH(
    Tree& d_tree,
    const thrust::device_vector<TPrimitive>& d_primitives,
    const thrust::device_vector<DeltaType>& d_deltas,
    const AABBFunc AABB,
    const bool wipe = false)
{
    build_ALBVH(d_tree, detail::raw(d_primitives), detail::raw(d_deltas), thrust::less<DeltaType>(),
                AABB, wipe);
}

} // namespace grace
