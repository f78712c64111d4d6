// grace/generic/meta.h -- vector-type <-> scalar-type maps (reference
// include/grace/generic/meta.h:7-90), as used by tests/helper/tree.cuh:19,35.
#pragma once

#include "grace/types.h"

namespace grace {

template <typename> struct Real2ToRealMapper;
template <> struct Real2ToRealMapper<float2> { typedef float type; };
template <> struct Real2ToRealMapper<double2> { typedef double type; };

template <typename> struct Real3ToRealMapper;
template <> struct Real3ToRealMapper<float3> { typedef float type; };
template <> struct Real3ToRealMapper<double3> { typedef double type; };

template <typename> struct Real4ToRealMapper;
template <> struct Real4ToRealMapper<float4> { typedef float type; };
template <> struct Real4ToRealMapper<double4> { typedef double type; };

template <typename> struct RealToReal2Mapper;
template <> struct RealToReal2Mapper<float> { typedef float2 type; };
template <> struct RealToReal2Mapper<double> { typedef double2 type; };

template <typename> struct RealToReal3Mapper;
template <> struct RealToReal3Mapper<float> { typedef float3 type; };
template <> struct RealToReal3Mapper<double> { typedef double3 type; };

template <typename> struct RealToReal4Mapper;
template <> struct RealToReal4Mapper<float> { typedef float4 type; };
template <> struct RealToReal4Mapper<double> { typedef double4 type; };

} // namespace grace
