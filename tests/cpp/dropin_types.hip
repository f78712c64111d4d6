// Drop-in check of the remaining template instantiations of the reference's API, through the
// reference's include paths over thrust::device_vector (authored here):
//   (1) float4 spheres, 63-bit keys: morton_keys63_sort_sph -> XOR_deltas_sph<uinteger64> ->
//       ALBVH_sph<float4, uinteger64> (build_sph.cuh:65-82,108-124), surface_area_deltas_sph;
//   (2) double4 spheres end to end: build_tree<double4> (double deltas), trace_hitcounts_sph,
//       trace_cumulative_sph<double4, double>, trace_sph<double4, int, double>,
//       sort_by_distance<double, int, double>, exclusive_segmented_scan<double>,
//       trace_with_sentinels_sph.
// Every traced quantity is compared with a host brute-force loop over grace::sphere_hit in the
// same precision.  Exit code 0 = PASSED.
#include "grace/cuda/build_sph.cuh"
#include "grace/cuda/functors/trace.cuh"
#include "grace/cuda/gen_rays.cuh"
#include "grace/cuda/kernels/bintree_trace.cuh"
#include "grace/cuda/nodes.h"
#include "grace/cuda/scan.cuh"
#include "grace/cuda/sort.cuh"
#include "grace/cuda/trace_sph.cuh"
#include "grace/cuda/device/intersect.cuh"
#include "helper/tree.cuh"

#include <thrust/device_vector.h>
#include <thrust/host_vector.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>

// OnHit_sphere_cumulate in double on the host (functors/trace.cuh:164-186, interpolate.h:11-39).
static double integral_d(double b2, double w)
{
    const double ir = 1.f / w;
    double x = (grace::N_table - 1) * (std::sqrt(b2) * ir);
    int i = int(x);
    if (i >= grace::N_table - 1) { x = grace::N_table - 1; i = grace::N_table - 2; }
    const double* t = grace::KernelIntegrals<double>::table;
    return std::fma(x - i, t[i + 1] - t[i], t[i]) * (ir * ir);
}

int main(int argc, char* argv[])
{
    size_t N = 60000, N_rays = 32 * 40;
    if (argc > 1) N = (size_t)std::strtol(argv[1], NULL, 10);
    if (argc > 2) N_rays = 32 * (size_t)std::strtol(argv[2], NULL, 10);
    int failures = 0;
    thrust::device_vector<grace::Ray> d_rays(N_rays);
    grace::uniform_random_rays(d_rays, 0.5f, 0.5f, 0.5f, 2.f, 77ull);
    thrust::host_vector<grace::Ray> h_rays = d_rays;

    // ---- (1) float4, 63-bit keys, XOR deltas
    {
        thrust::device_vector<float4> d_spheres;
        random_real4(make_float4(0.f, 0.f, 0.f, 0.005f), make_float4(1.f, 1.f, 1.f, 0.03f), N, d_spheres);
        const float3 bot = make_float3(0.f, 0.f, 0.f), top = make_float3(1.f, 1.f, 1.f);
        thrust::device_vector<grace::uinteger64> d_keys(N), d_deltas(N + 1);
        grace::morton_keys63_sort_sph(d_spheres, bot, top);
        grace::morton_keys_sph(d_spheres, bot, top, d_keys);          // keys of the sorted spheres
        thrust::host_vector<grace::uinteger64> h_keys = d_keys;
        for (size_t i = 1; i < N; ++i) if (h_keys[i - 1] > h_keys[i]) { ++failures; break; }
        grace::XOR_deltas_sph(d_keys, d_deltas);
        grace::Tree d_tree(N, 16);
        grace::ALBVH_sph(d_spheres, d_deltas, d_tree);
        thrust::device_vector<int> d_counts(N_rays);
        grace::trace_hitcounts_sph(d_rays, d_spheres, d_tree, d_counts);
        thrust::host_vector<int> h_counts = d_counts;
        thrust::host_vector<float4> h_spheres = d_spheres;
        for (size_t r = 0; r < N_rays; ++r) {
            int hits = 0; float b2, d;
            for (size_t s = 0; s < N; ++s) if (grace::sphere_hit(h_rays[r], h_spheres[s], b2, d)) ++hits;
            if (hits != h_counts[r]) ++failures;
        }
        // surface-area deltas build the same kind of tree: counts must not change
        thrust::device_vector<float> d_sa(N + 1);
        grace::surface_area_deltas_sph(d_spheres, d_sa);
        grace::Tree d_tree2(N, 16);
        grace::ALBVH_sph(d_spheres, d_sa, d_tree2);
        thrust::device_vector<int> d_counts2(N_rays);
        grace::trace_hitcounts_sph(d_rays, d_spheres, d_tree2, d_counts2);
        thrust::host_vector<int> h_counts2 = d_counts2;
        for (size_t r = 0; r < N_rays; ++r) if (h_counts2[r] != h_counts[r]) ++failures;
        std::cout << "float4 / 63-bit keys / XOR + area deltas: " << (failures ? "FAILED" : "ok") << std::endl;
    }

    // ---- (2) double4 end to end
    {
        thrust::device_vector<double4> d_spheres;
        random_real4(make_double4(0., 0., 0., 0.005), make_double4(1., 1., 1., 0.03), N, d_spheres);
        grace::Tree d_tree(N, 32);
        build_tree(d_spheres, make_double3(0., 0., 0.), make_double3(1., 1., 1.), d_tree);
        thrust::host_vector<double4> h_spheres = d_spheres;

        thrust::device_vector<int> d_counts(N_rays);
        grace::trace_hitcounts_sph(d_rays, d_spheres, d_tree, d_counts);
        thrust::device_vector<double> d_cum(N_rays);
        grace::trace_cumulative_sph(d_rays, d_spheres, d_tree, d_cum);
        thrust::device_vector<int> d_offsets(N_rays);
        thrust::device_vector<int> d_idx;
        thrust::device_vector<double> d_int, d_dist;
        grace::trace_sph(d_rays, d_spheres, d_tree, d_offsets, d_idx, d_int, d_dist);

        thrust::host_vector<int> h_counts = d_counts, h_offsets = d_offsets, h_idx = d_idx;
        thrust::host_vector<double> h_cum = d_cum, h_int = d_int, h_dist = d_dist;
        int before = failures;
        size_t at = 0;
        for (size_t r = 0; r < N_rays; ++r) {
            int hits = 0; double b2, d, cls[8] = { 0., 0., 0., 0., 0., 0., 0., 0. };
            if (h_offsets[r] != (int)at) ++failures;
            for (size_t s = 0; s < N; ++s)
                if (grace::sphere_hit(h_rays[r], h_spheres[s], b2, d)) {
                    const double w = integral_d(b2, h_spheres[s].w);
                    cls[(s >> 10) & 7] += w;      // the library's summation order: 8 interleaved classes
                    if (at < h_idx.size() && (h_idx[at] != (int)s || h_int[at] != w || h_dist[at] != d)) ++failures;
                    ++at; ++hits;
                }
            const double sum = ((cls[0] + cls[1]) + (cls[2] + cls[3])) + ((cls[4] + cls[5]) + (cls[6] + cls[7]));
            if (hits != h_counts[r] || sum != h_cum[r]) ++failures;
        }
        if (at != h_idx.size()) ++failures;
        std::cout << "double4 hit counts / column densities / per-hit outputs (bit-exact vs host): "
                  << (failures == before ? "ok" : "FAILED") << std::endl;

        before = failures;
        grace::sort_by_distance(d_dist, d_offsets, d_idx, d_int);
        thrust::device_vector<double> d_depth(d_int.size());
        grace::exclusive_segmented_scan(d_offsets, d_int, d_depth);
        h_dist = d_dist; h_int = d_int;
        thrust::host_vector<double> h_depth = d_depth;
        for (size_t r = 0; r < N_rays; ++r) {
            const size_t b = h_offsets[r], e = r + 1 < N_rays ? (size_t)h_offsets[r + 1] : h_dist.size();
            double run = 0.0;
            for (size_t k = b; k < e; ++k) {
                if (k > b && h_dist[k] < h_dist[k - 1]) ++failures;
                if (std::fabs(h_depth[k] - run) > 1e-12 * (1.0 + std::fabs(run))) ++failures;
                run += h_int[k];
            }
        }
        std::cout << "double sort_by_distance + exclusive_segmented_scan: " << (failures == before ? "ok" : "FAILED") << std::endl;

        before = failures;
        thrust::device_vector<int> d_off2(N_rays), d_idx2;
        thrust::device_vector<double> d_int2, d_dist2;
        grace::trace_with_sentinels_sph(d_rays, d_spheres, d_tree, d_off2, d_idx2, -7, d_int2, -1.5, d_dist2, 1e30);
        thrust::host_vector<int> h_off2 = d_off2, h_idx2 = d_idx2;
        thrust::host_vector<double> h_int2 = d_int2, h_dist2 = d_dist2;
        if (h_idx2.size() != h_idx.size() + N_rays) ++failures;
        for (size_t r = 0; r < N_rays && failures == before; ++r) {
            const size_t e = (size_t)h_off2[r] + h_counts[r];       // the ray's sentinel slot
            if (h_off2[r] != h_offsets[r] + (int)r) ++failures;
            if (h_idx2[e] != -7 || h_int2[e] != -1.5 || h_dist2[e] != 1e30) ++failures;
        }
        std::cout << "double trace_with_sentinels_sph: " << (failures == before ? "ok" : "FAILED") << std::endl;
    }
    // ---- (3) the generic functor trace composed like trace_sph's second pass
    //          (trace_sph.cuh:143-167: RayEntry_from_array + OnHit_sphere_individual) must
    //          reproduce the library's per-hit outputs bit for bit
    {
        const int before = failures;
        thrust::device_vector<float4> d_spheres;
        random_real4(make_float4(0.f, 0.f, 0.f, 0.005f), make_float4(1.f, 1.f, 1.f, 0.03f), N, d_spheres);
        grace::Tree d_tree(N, 32);
        build_tree(d_spheres, make_float3(0.f, 0.f, 0.f), make_float3(1.f, 1.f, 1.f), d_tree);
        thrust::device_vector<int> d_offsets(N_rays);
        thrust::device_vector<int> d_idx;
        thrust::device_vector<float> d_int, d_dist;
        grace::trace_sph(d_rays, d_spheres, d_tree, d_offsets, d_idx, d_int, d_dist);

        thrust::device_vector<int> g_idx(d_idx.size());
        thrust::device_vector<float> g_int(d_idx.size()), g_dist(d_idx.size());
        const double* p_table = &(grace::KernelIntegrals<double>::table[0]);
        thrust::device_vector<double> d_lookup(p_table, p_table + grace::N_table);
        typedef grace::RayData_sphere<int, float> RayData;
        grace::trace_texref<RayData>(
            d_rays, d_spheres, d_tree, sizeof(double) * grace::N_table,
            grace::InitGlobalToSmem<double>(thrust::raw_pointer_cast(d_lookup.data()), grace::N_table),
            grace::Intersect_sphere_b2dist(),
            grace::OnHit_sphere_individual<int, float>(thrust::raw_pointer_cast(g_idx.data()),
                                                       thrust::raw_pointer_cast(g_int.data()),
                                                       thrust::raw_pointer_cast(g_dist.data()),
                                                       grace::N_table),
            grace::RayEntry_from_array<int>(thrust::raw_pointer_cast(d_offsets.data())),
            grace::RayExit_null());
        thrust::host_vector<int> a = d_idx, b = g_idx;
        thrust::host_vector<float> aw = d_int, bw = g_int, ad = d_dist, bd = g_dist;
        for (size_t k = 0; k < a.size(); ++k)
            if (a[k] != b[k] || aw[k] != bw[k] || ad[k] != bd[k]) { ++failures; break; }
        if (a.size() == 0) ++failures;
        std::cout << "generic trace_texref + OnHit_sphere_individual == trace_sph (" << a.size()
                  << " hits): " << (failures == before ? "ok" : "FAILED") << std::endl;
    }
    std::cout << (failures ? "FAILED" : "PASSED") << std::endl;
    return failures ? EXIT_FAILURE : EXIT_SUCCESS;
}
