// Drop-in check, authored here (not the reference's file): the call sequence of
// tests/tree_traversal/tree_traversal.cu:40-100 -- random_spheres_tree, uniform_random_rays,
// trace_hitcounts_sph, host brute force with grace::sphere_hit, per-ray comparison -- against
// the REFERENCE's include paths and thrust::device_vector types, compiled by hipcc
// (-ffp-contract=off, the reference's -fmad=false) and linked with libgrace_hip.so.
//   dropin_tree_traversal <N> <N_rays/32> <max_per_leaf> <out.i32>
// Also exercises trace_sph + sort_by_distance + exclusive_segmented_scan (tests/distance_sort,
// tests/segmented_scan call sites).  Exit code 0 = PASSED.
#include "grace/cuda/nodes.h"
#include "grace/cuda/gen_rays.cuh"
#include "grace/cuda/scan.cuh"
#include "grace/cuda/sort.cuh"
#include "grace/cuda/trace_sph.cuh"
#include "grace/cuda/device/intersect.cuh"
#include "grace/ray.h"
#include "helper/tree.cuh"

#include <thrust/device_vector.h>
#include <thrust/host_vector.h>

#include <cstdio>
#include <cstdlib>
#include <iostream>

int main(int argc, char* argv[])
{
    size_t N = 200000;
    size_t N_rays = 32 * 100;
    int max_per_leaf = 32;
    if (argc > 1) N = (size_t)std::strtol(argv[1], NULL, 10);
    if (argc > 2) N_rays = 32 * (size_t)std::strtol(argv[2], NULL, 10);
    if (argc > 3) max_per_leaf = (int)std::strtol(argv[3], NULL, 10);

    thrust::device_vector<float4> d_spheres(N);
    thrust::device_vector<grace::Ray> d_rays(N_rays);
    thrust::device_vector<int> d_hit_counts(N_rays);
    grace::Tree d_tree(N, max_per_leaf);

    float4 low = make_float4(-1E4f, -1E4f, -1E4f, 80.f);
    float4 high = make_float4(1E4f, 1E4f, 1E4f, 400.f);
    random_spheres_tree(low, high, N, d_spheres, d_tree);
    grace::uniform_random_rays(d_rays, 0.f, 0.f, 0.f, 2E4f);

    grace::trace_hitcounts_sph(d_rays, d_spheres, d_tree, d_hit_counts);

    thrust::host_vector<float4> h_spheres = d_spheres;
    thrust::host_vector<grace::Ray> h_rays = d_rays;
    thrust::host_vector<int> h_hit_counts = d_hit_counts;
    size_t failed_rays = 0;
    double total = 0;
    for (size_t ri = 0; ri < N_rays; ++ri) {
        grace::Ray ray = h_rays[ri];
        int hits = 0;
        float b2, d;
        for (size_t si = 0; si < N; ++si)
            if (grace::sphere_hit(ray, h_spheres[si], b2, d)) ++hits;
        if (hits != h_hit_counts[ri]) ++failed_rays;
        total += hits;
    }
    std::cout << "Mean of " << total / N_rays << " hits per ray (host)." << std::endl;

    // per-hit outputs, per-ray sort by distance, optical depth in front of every hit
    thrust::device_vector<int> d_ray_offsets(N_rays);
    thrust::device_vector<unsigned int> d_hit_indices;
    thrust::device_vector<float> d_hit_integrals, d_hit_distances;
    grace::trace_sph(d_rays, d_spheres, d_tree, d_ray_offsets, d_hit_indices, d_hit_integrals,
                     d_hit_distances);
    grace::sort_by_distance(d_hit_distances, d_ray_offsets, d_hit_indices, d_hit_integrals);
    thrust::device_vector<float> d_depth(d_hit_integrals.size());
    grace::exclusive_segmented_scan(d_ray_offsets, d_hit_integrals, d_depth);
    thrust::host_vector<int> h_offsets = d_ray_offsets;
    thrust::host_vector<float> h_dist = d_hit_distances;
    size_t unsorted = 0;
    if (d_hit_distances.size() != (size_t)total) ++failed_rays;
    for (size_t ri = 0; ri < N_rays; ++ri) {
        const size_t b = h_offsets[ri], e = ri + 1 < N_rays ? h_offsets[ri + 1] : h_dist.size();
        for (size_t k = b + 1; k < e; ++k)
            if (h_dist[k] < h_dist[k - 1] || h_dist[k] < 0) ++unsorted;
    }

    if (argc > 4) {
        std::FILE* f = std::fopen(argv[4], "wb");
        if (!f || std::fwrite(thrust::raw_pointer_cast(h_hit_counts.data()), sizeof(int), N_rays, f) != N_rays) return 3;
        std::fclose(f);
    }
    if (failed_rays == 0 && unsorted == 0) { std::cout << "PASSED" << std::endl; return EXIT_SUCCESS; }
    std::cout << "FAILED: " << failed_rays << " rays, " << unsorted << " unsorted hits" << std::endl;
    return EXIT_FAILURE;
}
