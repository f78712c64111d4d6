// profile_trace_gadget -- per-call trace times on a Gadget-2 snapshot.  Mirror of the
// reference's tests/profile_trace_gadget/profile_trace_gadget.cu (same arguments and output
// lines) on the drop-in header.  "synthetic:<N>" in place of a file name generates N uniform
// particles with the 48-neighbour smoothing length instead of reading a snapshot.
//
//   profile_trace_gadget [N_rays/32 [max_per_leaf [file [N_iter]]]]
#include "grace/grace.h"
#include "grace/read_gadget.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

namespace {
struct Timer {   // tests/helper/cuda_timer.cuh: split() = ms since the previous split
    std::chrono::steady_clock::time_point t0, last;
    static void sync() { grace::detail::check(grace_stream_synchronize(nullptr)); }
    void start() { sync(); t0 = last = std::chrono::steady_clock::now(); }
    double split()
    {
        sync();
        const auto now = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(now - last).count();
        last = now;
        return ms;
    }
    double elapsed()
    {
        sync();
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
};
} // namespace

int main(int argc, char* argv[])
{
    std::cout.setf(std::ios::fixed, std::ios::floatfield);
    std::cout.precision(3);

    size_t N_rays = 1200 * 32;
    int max_per_leaf = 32;
    std::string fname = "../data/gadget/0128/Data_025";
    int N_iter = 2;
    if (argc > 1) N_rays = 32 * size_t(std::strtol(argv[1], NULL, 10));
    if (argc > 2) max_per_leaf = int(std::strtol(argv[2], NULL, 10));
    if (argc > 3) fname = std::string(argv[3]);
    if (argc > 4) N_iter = int(std::strtol(argv[4], NULL, 10));

    std::cout << "Gadget file:            " << fname << std::endl;
    std::vector<grace::float4> h_spheres;
    if (fname.rfind("synthetic:", 0) == 0) {
        const size_t n = size_t(std::strtol(fname.c_str() + 10, NULL, 10));
        h_spheres.resize(n);
        const float h = float(std::cbrt(3.0 * 48.0 / (4.0 * 3.141592653589793 * n)));
        uint64_t s = 42;
        auto u = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull;
                         return float((s >> 40) * (1.0 / 16777216.0)); };
        for (size_t i = 0; i < n; ++i) h_spheres[i] = grace::make_float4(u(), u(), u(), h);
    } else {
        read_gadget(fname, h_spheres);
    }
    grace::device_vector<grace::float4> d_spheres(h_spheres);
    const size_t N = d_spheres.size();

    std::cout << "Number of particles:    " << N << std::endl
              << "Number of rays:         " << N_rays << std::endl
              << "Max particles per leaf: " << max_per_leaf << std::endl
              << "Number of iterations:   " << N_iter << std::endl
              << "Running on device:      0 (AMD Instinct, libgrace_hip " << grace_version() << ")"
              << std::endl << std::endl;

    grace::Tree d_tree(N, max_per_leaf);
    build_tree(d_spheres, d_tree);

    // Ray origin is the box centre; all rays exit the box (profile_trace_gadget.cu:78-84).
    grace::float4 mins, maxs;
    grace::min_max_vec4(d_spheres, &mins, &maxs);
    const float origin = float((maxs.x + mins.x) / 2.);
    const float length = 2 * (maxs.x - mins.x);

    Timer timer;
    double t_genray = 0, t_sort = 0, t_cum = 0, t_trace = 0, t_hit = 0, t_all = 0;
    for (int i = -1; i < N_iter; ++i) {
        timer.start();
        grace::device_vector<grace::Ray> d_rays(N_rays);
        grace::device_vector<int> d_ray_offsets(N_rays);
        grace::device_vector<float> d_integrals(N_rays);
        grace::device_vector<int> d_indices;
        grace::device_vector<float> d_distances;
        timer.split();

        grace::uniform_random_rays(d_rays, origin, origin, origin, length);
        if (i >= 0) t_genray += timer.split();

        grace::trace_cumulative_sph(d_rays, d_spheres, d_tree, d_integrals);
        if (i >= 0) t_cum += timer.split();

        grace::trace_sph(d_rays, d_spheres, d_tree, d_ray_offsets, d_indices, d_integrals, d_distances);
        if (i >= 0) t_trace += timer.split();

        grace::sort_by_distance(d_distances, d_ray_offsets, d_indices, d_integrals);
        if (i >= 0) t_sort += timer.split();

        grace::trace_hitcounts_sph(d_rays, d_spheres, d_tree, d_ray_offsets);
        if (i >= 0) t_hit += timer.split();

        if (i >= 0) t_all += timer.elapsed();

        if (i == 0) {
            double trace_bytes = 0.0;
            trace_bytes += d_spheres.size() * sizeof(grace::float4);
            trace_bytes += d_tree.leaves.size() * sizeof(grace::int4);
            trace_bytes += d_tree.nodes.size() * sizeof(grace::int4);
            trace_bytes += d_rays.size() * sizeof(grace::Ray);
            trace_bytes += d_ray_offsets.size() * sizeof(int);
            trace_bytes += d_integrals.size() * sizeof(float);
            trace_bytes += d_indices.size() * sizeof(int);
            trace_bytes += d_distances.size() * sizeof(float);
            trace_bytes += 51 * sizeof(double); // integral lookup table
            std::cout << "Total hits: " << d_indices.size() << std::endl << std::endl
                      << "Total memory for full trace kernel and sort: "
                      << trace_bytes / (1024.0 * 1024.0 * 1024.0) << " GiB" << std::endl << std::endl;
        }
    }

    std::cout << "Time for generating and sorting rays:   " << std::setw(8) << t_genray / N_iter << " ms" << std::endl
              << "Time for hit count tracing:             " << std::setw(8) << t_hit / N_iter << " ms" << std::endl
              << "Time for cumulative density tracing:    " << std::setw(8) << t_cum / N_iter << " ms" << std::endl
              << "Time for full tracing:                  " << std::setw(8) << t_trace / N_iter << " ms" << std::endl
              << "Time for sort-by-distance:              " << std::setw(8) << t_sort / N_iter << " ms" << std::endl
              << "Time for total (inc. memory ops):       " << std::setw(8) << t_all / N_iter << " ms" << std::endl
              << std::endl;
    return EXIT_SUCCESS;
}
