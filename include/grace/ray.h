// grace/ray.h -- the reference's ray record (include/grace/ray.h:5-10): 28 bytes, direction
// (normalised) first, then origin, then length.  libgrace_hip.so reads exactly this layout.
#pragma once

namespace grace {

struct Ray
{
    float dx, dy, dz;
    float ox, oy, oz;
    float length;
};

} // namespace grace
