// helper/read_gadget.cuh -- Gadget-2 snapshot -> device vector of spheres, the reference tests'
// read_gadget(fname, d_spheres) (tests/helper/read_gadget.cuh:69-159) over grace/read_gadget.h.
#pragma once

#include "grace/read_gadget.h"

#include <thrust/device_vector.h>
#include <thrust/host_vector.h>

#include <vector>

inline void read_gadget(const std::string& fname, thrust::device_vector<float4>& d_spheres)
{
    std::vector<float4> h;
    read_gadget(fname, h);
    thrust::host_vector<float4> hv(h.begin(), h.end());
    d_spheres = hv;
}
