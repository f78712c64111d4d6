// grace/generic/morton.h -- grace::morton_key, host- and device-callable (reference
// include/grace/generic/morton.h:14-55): 30-bit keys from three 10-bit integers or three floats
// in (0, 1), 63-bit keys from three 21-bit integers or three doubles in (0, 1); x is the least
// significant dimension.  BASELINE config 1 (tests/morton_key: the CPU host path) runs on these;
// the GPU key kernels of libgrace_hip.so (csrc/morton.hip) call the same functions.
#pragma once

#include "grace/generic/bits.h"

namespace grace {

// 30-bit keys.
GRACE_HOST_DEVICE uinteger32 morton_key(const uinteger32 x, const uinteger32 y, const uinteger32 z)
{
    return detail::space_by_two_10bit(z) << 2 | detail::space_by_two_10bit(y) << 1
        | detail::space_by_two_10bit(x);
}

// 63-bit keys.
GRACE_HOST_DEVICE uinteger64 morton_key(const uinteger64 x, const uinteger64 y, const uinteger64 z)
{
    return detail::space_by_two_21bit(z) << 2 | detail::space_by_two_21bit(y) << 1
        | detail::space_by_two_21bit(x);
}

// 30-bit keys from floats, which must lie in (0, 1): each is scaled to [0, 1023) and truncated
// (morton.h:33-43; the product is formed in float, as there).
GRACE_HOST_DEVICE uinteger32 morton_key(const float x, const float y, const float z)
{
    const unsigned int span = (1u << 10) - 1;
    return morton_key(static_cast<uinteger32>(span * x), static_cast<uinteger32>(span * y),
                      static_cast<uinteger32>(span * z));
}

// 63-bit keys from doubles, which must lie in (0, 1) (morton.h:45-55).
GRACE_HOST_DEVICE uinteger64 morton_key(const double x, const double y, const double z)
{
    const unsigned int span = (1u << 21) - 1;
    return morton_key(static_cast<uinteger64>(span * x), static_cast<uinteger64>(span * y),
                      static_cast<uinteger64>(span * z));
}

} // namespace grace
