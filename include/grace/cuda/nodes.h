// grace/cuda/nodes.h -- grace::Tree, the BVH container of the reference
// (include/grace/cuda/nodes.h:14-71), over thrust::device_vector (rocThrust as the container).
//
// Layout (nodes.h:17-42), read and written as is by libgrace_hip.so:
//   nodes[4 i + 0] = {left child, right child, first leaf, last leaf}
//   nodes[4 i + 1] = {left  bx, tx, by, ty}      (float bits)
//   nodes[4 i + 2] = {right bx, tx, by, ty}
//   nodes[4 i + 3] = {left bz, tz, right bz, tz}
//   child index >= number of nodes  <=>  leaf (index - number of nodes)
//   leaves[j] = {first primitive, number of primitives, 0, 0}
// Both vectors are allocated for N leaves and shrunk by the build (albvh.cuh:842-845).
#pragma once

#include "grace/error.h"
#include "grace/types.h"

#include <thrust/device_vector.h>
#include <thrust/host_vector.h>

namespace grace {

class Tree
{
public:
    thrust::device_vector<int4> nodes;
    thrust::device_vector<int4> leaves;
    // A pointer to the *value of the index* of the root element of the tree (device memory).
    int* root_index_ptr;
    int max_per_leaf;

    Tree(size_t N_leaves, int max_per_leaf = 1) :
        nodes(4 * (N_leaves - 1)), leaves(N_leaves), root_index_ptr(NULL),
        max_per_leaf(max_per_leaf)
    {
        GRACE_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&root_index_ptr), sizeof(int)));
    }

    ~Tree()
    {
        GRACE_HIP_CHECK(hipFree(root_index_ptr));
    }

private:
    Tree(const Tree&);              // owns device memory: not copyable (nor is the reference's,
    Tree& operator=(const Tree&);   // whose copy would double-free root_index_ptr)
};

class H_Tree
{
public:
    thrust::host_vector<int4> nodes;
    thrust::host_vector<int4> leaves;
    int root_index;
    int max_per_leaf;

    H_Tree(size_t N_leaves, int _max_per_leaf = 1) :
        nodes(4 * (N_leaves - 1)), leaves(N_leaves), root_index(0),
        max_per_leaf(_max_per_leaf) {}
};

// nodes.h:77-87: a node's right child can never be node 0, and a leaf can never cover zero
// elements.
struct is_empty_node
{
    GRACE_HOST_DEVICE bool operator()(const int4 node) const { return node.y == 0; }
};

} // namespace grace
