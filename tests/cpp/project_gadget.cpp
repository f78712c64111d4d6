// Mirror of the reference's tests/project_gadget/project_gadget.cu (with the per-phase timing
// of tests/profile_project_gadget): read a Gadget-2 snapshot (or synthesise one), build the
// tree, trace a grid of orthographic -z rays with trace_cumulative_sph, print mean/max/min
// and write log10 column density as density.bmp.
//   project_gadget [N_rays/32] [max_per_leaf] [gadget file | "synthetic:N"] [out.bmp]
#include "grace/grace.h"
#include "grace/images.h"
#include "grace/read_gadget.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

static double now_ms()
{
    grace::detail::check(grace_stream_synchronize(nullptr));
    return std::chrono::duration<double, std::milli>(
               std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char* argv[])
{
    size_t N_rays = 512 * 512;
    int max_per_leaf = 32;
    std::string fname = "synthetic:200000";
    std::string out = "density.bmp";
    if (argc > 1) N_rays = 32 * (size_t)std::strtol(argv[1], NULL, 10);
    if (argc > 2) max_per_leaf = (int)std::strtol(argv[2], NULL, 10);
    if (argc > 3) fname = argv[3];
    if (argc > 4) out = argv[4];

    size_t N_per_side = (size_t)std::floor(std::pow((double)N_rays, 0.500001));
    N_per_side = ((N_per_side + 32 - 1) / 32) * 32;   // N_rays must be a multiple of 32
    N_rays = N_per_side * N_per_side;

    std::vector<grace::float4> h_spheres;
    if (fname.rfind("synthetic:", 0) == 0) {
        const size_t n = (size_t)std::strtol(fname.c_str() + 10, NULL, 10);
        h_spheres.resize(n);
        const float h = (float)std::cbrt(3.0 * 48.0 / (4.0 * 3.141592653589793 * n));
        uint64_t s = 42;
        auto u = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull;
                         return (float)((s >> 40) * (1.0 / 16777216.0)); };
        for (size_t i = 0; i < n; ++i) h_spheres[i] = grace::make_float4(u(), u(), u(), h);
    } else {
        read_gadget(fname, h_spheres);
    }
    const size_t N = h_spheres.size();
    std::printf("Gadget file:             %s\nNumber of particles:     %zu\nNumber of rays:          %zu\n"
                "Number of rays per side: %zu\nMax particles per leaf:  %d\n\n",
                fname.c_str(), N, N_rays, N_per_side, max_per_leaf);

    grace::device_vector<grace::float4> d_spheres(h_spheres);
    grace::device_vector<grace::Ray> d_rays(N_rays);
    grace::Tree d_tree(N, max_per_leaf);

    grace::float4 mins, maxs;
    grace::min_max_vec4(d_spheres, &mins, &maxs);
    mins.w = maxs.w = 0;      // project_gadget.cu:69-71: no padding for images

    const double t0 = now_ms();
    build_tree(d_spheres, mins, maxs, d_tree);
    const double t1 = now_ms();
    orthogonal_rays_z(N_per_side, mins, maxs, d_rays);
    const double t2 = now_ms();
    grace::device_vector<float> d_integrals(N_rays);
    grace::trace_cumulative_sph(d_rays, d_spheres, d_tree, d_integrals);
    const double t3 = now_ms();

    std::vector<float> h = d_integrals.to_host();
    float mx = 0.f, mn = 1e20f; double sum = 0;
    for (size_t i = 0; i < h.size(); ++i) { mx = std::max(mx, h[i]); mn = std::min(mn, h[i]); sum += h[i]; }
    std::printf("Time for tree build:     %.3f ms\nTime for ray generation: %.3f ms\n"
                "Time for tracing:        %.3f ms (%.1f Mrays/s)\n\n", t1 - t0, t2 - t1, t3 - t2,
                N_rays / (t3 - t2) / 1e3);
    std::printf("Mean output %g\nMax output: %g\nMin output: %g\n\n", sum / h.size(), mx, mn);

    mn = std::max(1E-20f, mn);
    for (size_t i = 0; i < h.size(); ++i) h[i] = std::log10(h[i]);
    make_bitmap(h.data(), N_per_side, N_per_side, std::log10(mn), std::log10(mx), out);
    std::printf("Wrote %s\n", out.c_str());
    return EXIT_SUCCESS;
}
