// grace/detail/config.h -- function qualifiers and integer typedefs shared by the drop-in header
// set (grace/types.h) and by the headers that are also usable from plain host C++
// (grace/generic/bits.h, morton.h: the reference's host-callable morton_key).  Under hipcc the
// qualifiers are the reference's (include/grace/types.h:14-32); under a host-only compiler they
// reduce to `inline`.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GRACE_HOST __host__ inline
#define GRACE_DEVICE __device__ inline
#define GRACE_HOST_DEVICE __host__ __device__ inline
#else
#define GRACE_HOST inline
#define GRACE_DEVICE inline
#define GRACE_HOST_DEVICE inline
#endif

namespace grace {

typedef uint32_t uinteger32;
typedef uint64_t uinteger64;
typedef int32_t integer32;
typedef int64_t integer64;

} // namespace grace
