// grace/generic/bits.h -- sgn() and the bit-spreading helpers behind the Morton keys
// (reference include/grace/generic/bits.h:11-46), callable on the host and on the device.
//
// space_by_two_10bit(x): the low 10 bits of x with two zero bits between neighbours (bit i moves
// to bit 3 i), 30 bits in all; space_by_two_21bit: the same for 21 bits / 63 bits.  Each step of
// the classic shift-or-mask ladder doubles the gap between groups of bits that are already in
// place; the masks keep exactly the bits that belong there, so the result is fixed by the
// definition (and pinned by the reference's known-answer tests, tests/morton_key/30bit_key.cu:
// 20-26, 63bit_key.cu:20-26 -- tests/golden/kat.json).  libgrace_hip.so's key kernels
// (csrc/morton.hip) use these same functions.
#pragma once

#include "grace/detail/config.h"

namespace grace {

template <typename T>
GRACE_HOST_DEVICE int sgn(T val)
{
    return (T(0) < val) - (val < T(0));
}

namespace detail {

template <typename UInteger>
GRACE_HOST_DEVICE uinteger32 space_by_two_10bit(const UInteger x)
{
    uinteger32 v = static_cast<uinteger32>(x) & 0x3FFu;   // ten bits: higher ones are dropped
    v = (v | (v << 16)) & 0x030000FFu;                    // 2 | 8
    v = (v | (v << 8)) & 0x0300F00Fu;                     // 2 | 4 | 4
    v = (v | (v << 4)) & 0x030C30C3u;                     // pairs
    v = (v | (v << 2)) & 0x09249249u;                     // single bits, every third position
    return v;
}

template <typename UInteger>
GRACE_HOST_DEVICE uinteger64 space_by_two_21bit(const UInteger x)
{
    uinteger64 v = static_cast<uinteger64>(x) & 0x1FFFFFull;   // twenty-one bits
    v = (v | (v << 32)) & 0x001F00000000FFFFull;               // 5 | 16
    v = (v | (v << 16)) & 0x001F0000FF0000FFull;               // 5 | 8 | 8
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;                // 1 | 4 | 4 ...
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;                // pairs
    v = (v | (v << 2)) & 0x1249249249249249ull;                // single bits, every third position
    return v;
}

} // namespace detail

} // namespace grace
