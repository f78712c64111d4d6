// profile_tree -- per-phase LBVH build times for 2^p random spheres, p = log2N_min .. log2N_max.
// Mirror of the reference's tests/profile_tree/profile_tree.cu (same arguments, same output
// lines, so that scrapers such as tests/profile_leafbuilders.py keep working) on the drop-in
// header.  Phases the reference runs as separate kernels and this library fuses are reported
// on the line of the phase that now carries them:
//   "building leaves"        = leaf heads + scan + leaf records + leaf deltas (fused)
//   "computing leaf deltas"  = 0 (part of the line above)
//   "building nodes"         = leaf boxes + pyramids + nodes
//
//   profile_tree [max_per_leaf [N_iter [log2N_max | log2N_min log2N_max]]]
#include "grace/grace.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iomanip>
#include <iostream>
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
uint64_t splitmix(uint64_t& s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
} // namespace

int main(int argc, char* argv[])
{
    std::cout.setf(std::ios::fixed, std::ios::floatfield);
    std::cout.precision(3);

    int max_per_leaf = 32, N_iter = 100, log2N_min = 20, log2N_max = 23;
    if (argc > 1) max_per_leaf = int(std::strtol(argv[1], NULL, 10));
    if (argc > 2) N_iter = int(std::strtol(argv[2], NULL, 10));
    if (argc > 3) {
        log2N_max = std::min(28, std::max(5, int(std::strtol(argv[3], NULL, 10))));
        if (log2N_max < log2N_min) log2N_min = log2N_max;
    }
    if (argc > 4) {
        log2N_max = std::min(28, std::max(5, int(std::strtol(argv[4], NULL, 10))));
        log2N_min = std::min(28, std::max(5, int(std::strtol(argv[3], NULL, 10))));
    }

    std::cout << "Max particles per leaf:   " << max_per_leaf << std::endl
              << "Iterations per tree:      " << N_iter << std::endl
              << "Starting log2(N_points):  " << log2N_min << std::endl
              << "Finishing log2(N_points): " << log2N_max << std::endl
              << "Running on device:        0 (AMD Instinct, libgrace_hip " << grace_version() << ")"
              << std::endl << std::endl;

    grace::detail::check(grace_albvh_enable_timing(1));
    for (int p = log2N_min; p <= log2N_max; ++p) {
        const size_t N = size_t(1) << p;
        const grace::float3 low = grace::make_float3(0.f, 0.f, 0.f), high = grace::make_float3(1.f, 1.f, 1.f);
        // centres U[0,1)^3, radii U[0,0.1) (profile_tree.cu:77-84); own generator
        std::vector<grace::float4> h_spheres(N);
        uint64_t seed = 0x5eed + p;
        for (size_t i = 0; i < N; ++i) {
            const uint64_t a = splitmix(seed), b = splitmix(seed);
            h_spheres[i].x = float((a >> 40) * (1.0 / 16777216.0));
            h_spheres[i].y = float(((a >> 16) & 0xFFFFFF) * (1.0 / 16777216.0));
            h_spheres[i].z = float((b >> 40) * (1.0 / 16777216.0));
            h_spheres[i].w = float(((b >> 16) & 0xFFFFFF) * (0.1 / 16777216.0));
        }

        Timer timer;
        double t_all = 0, t_morton = 0, t_sort = 0, t_deltas = 0, t_leaves = 0, t_leaf_deltas = 0, t_nodes = 0;
        for (int i = -1; i < N_iter; ++i) {
            timer.start();
            grace::device_vector<grace::float4> d_spheres(h_spheres);
            grace::device_vector<grace::uinteger32> d_keys(N);
            grace::device_vector<float> d_deltas(N + 1);
            grace::Tree d_tree(N, max_per_leaf);
            timer.split();   // allocations and the upload are not part of t_morton

            grace::morton_keys_sph(d_spheres, low, high, d_keys);
            if (i >= 0) t_morton += timer.split();

            grace::detail::check(grace_sort_pairs_u32(d_keys.data(), d_spheres.data(), N,
                                                      sizeof(grace::float4), 0, 30, nullptr, nullptr));
            if (i >= 0) t_sort += timer.split();

            grace::euclidean_deltas_sph(d_spheres, d_deltas);
            if (i >= 0) t_deltas += timer.split();

            grace::ALBVH_sph(d_spheres, d_deltas, d_tree);
            float ms_leaves = 0, ms_nodes = 0;
            grace::detail::check(grace_albvh_last_phase_ms(&ms_leaves, &ms_nodes));
            timer.split();
            if (i >= 0) { t_leaves += ms_leaves; t_nodes += ms_nodes; }

            if (i >= 0) t_all += timer.elapsed();
        }

        std::cout << "Number of particles:               " << N << std::endl
                  << "Time for Morton key generation:    " << std::setw(7) << t_morton / N_iter << " ms." << std::endl
                  << "Time for sort-by-key:              " << std::setw(7) << t_sort / N_iter << " ms." << std::endl
                  << "Time for computing deltas:         " << std::setw(7) << t_deltas / N_iter << " ms." << std::endl
                  << "Time for building leaves:          " << std::setw(7) << t_leaves / N_iter << " ms." << std::endl
                  << "Time for computing leaf deltas:    " << std::setw(7) << t_leaf_deltas / N_iter << " ms." << std::endl
                  << "Time for building nodes:           " << std::setw(7) << t_nodes / N_iter << " ms." << std::endl
                  << "Time for total (inc. memory ops):  " << std::setw(7) << t_all / N_iter << " ms." << std::endl
                  << std::endl;
    }
    return EXIT_SUCCESS;
}
