// grace/cuda/functors/trace.cuh -- the reference's stock traversal functors and ray payloads
// (include/grace/cuda/functors/trace.cuh:18-235, generic/raydata.h:5-16): Init_null,
// InitGlobalToSmem, RayEntry_null / _from_array, RayExit_null / _to_array,
// Intersect_sphere_bool / _b2dist, OnHit_increment / _sphere_cumulate / _sphere_individual,
// RayData_datum / _sphere.  They live with the generic kernel in grace/hip/trace_core.hpp.
#pragma once

#include "grace/hip/trace_core.hpp"
