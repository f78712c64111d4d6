// grace/cuda/device/intersect.cuh -- the include path through which the reference's callers
// reach grace::sphere_hit (tests/tree_traversal/tree_traversal.cu:10).  The reference's
// AABBs_hit (device/intersect.cuh:10-40) lives inside libgrace_hip.so's traversal kernel and in
// include/grace/hip/trace.hpp for user-functor kernels.
#pragma once

#include "grace/generic/intersect.h"
