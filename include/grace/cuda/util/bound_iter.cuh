// grace/cuda/util/bound_iter.cuh -- grace::gpu::BoundIter, the bounds-carrying view of a trace
// kernel's user LDS bytes that every trace functor receives (reference
// include/grace/cuda/util/bound_iter.cuh:18-229).  Defined with the generic traversal.
#pragma once
#include "grace/hip/trace_core.hpp"
