// Drop-in check, authored here (not the reference's file): a caller written against the
// REFERENCE's include paths and thrust::device_vector types -- the call sequence of
// tests/project_gadget/project_gadget.cu:58-96 (read snapshot, min/max, build_tree,
// orthogonal_rays_z, trace_cumulative_sph, reductions, log10, bitmap) -- compiled by hipcc
// against include/grace/cuda/*.cuh and linked with libgrace_hip.so.  Only <curand_kernel.h>, a
// CUDA-toolkit header the reference includes to work around a Thrust bug, is gone.
//   dropin_project_gadget <N_rays/32> <max_per_leaf> <gadget file> <out.f32> [out.bmp]
// Writes the raw float image to out.f32 so that the test can compare it bit for bit with the
// ctypes path.
#include "grace/cuda/build_sph.cuh"
#include "grace/cuda/nodes.h"
#include "grace/cuda/trace_sph.cuh"
#include "grace/cuda/util/extrema.cuh"
#include "grace/ray.h"
#include "helper/images.hpp"
#include "helper/tree.cuh"
#include "helper/rays.cuh"
#include "helper/read_gadget.cuh"

#include <thrust/device_vector.h>
#include <thrust/host_vector.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <iostream>
#include <string>

int main(int argc, char* argv[])
{
    if (argc < 5) { std::cerr << "usage: N_rays/32 max_per_leaf gadget_file out.f32 [out.bmp]\n"; return 2; }
    size_t N_rays = 32 * (size_t)std::strtol(argv[1], NULL, 10);
    int max_per_leaf = (int)std::strtol(argv[2], NULL, 10);
    std::string fname = argv[3];

    size_t N_per_side = std::floor(std::pow(N_rays, 0.500001));
    N_per_side = ((N_per_side + 32 - 1) / 32) * 32;     // N_rays must be a multiple of 32
    N_rays = N_per_side * N_per_side;

    thrust::device_vector<float4> d_spheres;             // resized in read_gadget()
    read_gadget(fname, d_spheres);
    const size_t N = d_spheres.size();
    std::cout << "Number of particles:     " << N << std::endl
              << "Number of rays:          " << N_rays << std::endl;

    thrust::device_vector<grace::Ray> d_rays(N_rays);
    grace::Tree d_tree(N, max_per_leaf);

    float4 mins, maxs;
    grace::min_vec4(d_spheres, &mins);
    grace::max_vec4(d_spheres, &maxs);
    mins.w = maxs.w = 0;

    build_tree(d_spheres, mins, maxs, d_tree);
    orthogonal_rays_z(N_per_side, mins, maxs, d_rays);

    thrust::device_vector<float> d_integrals(N_rays);
    grace::trace_cumulative_sph(d_rays, d_spheres, d_tree, d_integrals);

    // (the reference reduces on the device with thrust::reduce; a host loop here)
    thrust::host_vector<float> h_integrals = d_integrals;

    // The extensions: the same trace with the scene and the ray batch prepared (handles pin the
    // cached records while they live) -- same bits.
    {
        grace::PreparedTrace scene = grace::prepare_trace_sph(d_spheres, d_tree);
        grace::PreparedTrace rays = grace::prepare_trace_rays(d_rays);
        thrust::device_vector<float> d_again(N_rays);
        grace::trace_cumulative_sph(d_rays, d_spheres, d_tree, d_again);
        thrust::host_vector<float> h_again = d_again;
        if (std::memcmp(thrust::raw_pointer_cast(h_again.data()), thrust::raw_pointer_cast(h_integrals.data()),
                        N_rays * sizeof(float)) != 0) {
            std::cout << "prepared trace differs" << std::endl;
            return 4;
        }
    }
    float max_integral = 0.0f, min_integral = 1E20f;
    double sum = 0.0;
    for (size_t i = 0; i < N_rays; ++i) {
        max_integral = std::max(max_integral, h_integrals[i]);
        min_integral = std::min(min_integral, h_integrals[i]);
        sum += h_integrals[i];
    }
    std::cout << "Mean output " << sum / N_rays << std::endl
              << "Max output: " << max_integral << std::endl
              << "Min output: " << min_integral << std::endl;

    std::FILE* f = std::fopen(argv[4], "wb");
    if (!f || std::fwrite(thrust::raw_pointer_cast(h_integrals.data()), sizeof(float), N_rays, f) != N_rays) return 3;
    std::fclose(f);

    if (argc > 5) {
        min_integral = std::max(1E-20f, min_integral);
        for (size_t i = 0; i < N_rays; ++i) h_integrals[i] = std::log10(h_integrals[i]);
        make_bitmap(thrust::raw_pointer_cast(h_integrals.data()), N_per_side, N_per_side,
                    std::log10(min_integral), std::log10(max_integral), argv[5]);
    }
    return EXIT_SUCCESS;
}
