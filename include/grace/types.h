// grace/types.h -- integer typedefs, function qualifiers and the two enums of the reference's
// include/grace/types.h:14-51, for translation units compiled by hipcc for gfx950.
//
// Part of the drop-in header set (grace/cuda/build_sph.cuh, trace_sph.cuh, nodes.h, scan.cuh,
// sort.cuh, gen_rays.cuh, ray.h ...): the reference's include paths, names and template
// signatures over thrust::device_vector (rocThrust, used as the container only), every body a
// type dispatch onto the C ABI of libgrace_hip.so (include/grace_hip.h).  The HIP-free mirror
// include/grace/grace.h defines the same names over its own container; use one or the other
// in a translation unit, not both.
#pragma once

#ifdef GRACE_HIP_FREE_MIRROR_INCLUDED
#error "grace/grace.h (HIP-free mirror) and the grace/cuda/*.cuh drop-in headers define the same names: include one set only"
#endif
#define GRACE_DROPIN_HEADERS_INCLUDED 1

#include "grace/detail/config.h"   // GRACE_HOST / _DEVICE / _HOST_DEVICE, uinteger32/64, integer32/64

#include <hip/hip_runtime.h>   // float4, int4, double4, make_float3 ... (vector_types.h in the reference)

namespace grace {

// Binary encoding with +ve = 1, -ve = 0: octants for ray generation (types.h:36-45).
enum Octants { PPP = 7, PPM = 6, PMP = 5, PMM = 4, MPP = 3, MPM = 2, MMP = 1, MMM = 0 };

// types.h:47-51
enum RaySortType { NoSort, DirectionSort, EndPointSort };

} // namespace grace
