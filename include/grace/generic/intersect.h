// grace/generic/intersect.h -- grace::sphere_hit for host and device code (reference
// include/grace/generic/intersect.h:10-55; the tree_traversal test calls it on the host as its
// brute-force oracle, tests/tree_traversal/tree_traversal.cu:65-79).  Compile callers with
// -ffp-contract=off: the reference's CPU/GPU equality is stated for unfused arithmetic
// (tests/tree_traversal/Makefile:5-8), which is what libgrace_hip.so evaluates.
#pragma once

#include "grace/ray.h"
#include "grace/types.h"

namespace grace {

// All arithmetic in the precision of Real.  b2: squared impact parameter; dot_p: distance
// along the ray to the point of closest approach.  A ray that starts beyond the closest
// approach, or ends before it, misses.
template <typename Real4, typename Real>
GRACE_HOST_DEVICE bool sphere_hit(const Ray& ray, const Real4& sphere, Real& b2, Real& dot_p)
{
    const Real px = sphere.x - ray.ox, py = sphere.y - ray.oy, pz = sphere.z - ray.oz;
    const Real rx = ray.dx, ry = ray.dy, rz = ray.dz;
    dot_p = px * rx + py * ry + pz * rz;
    const Real bx = px - dot_p * rx, by = py - dot_p * ry, bz = pz - dot_p * rz;
    b2 = bx * bx + by * by + bz * bz;
    if (b2 >= sphere.w * sphere.w) return false;
    if (dot_p < 0.0f) return false;
    if (dot_p >= ray.length) return false;
    return true;
}

} // namespace grace
