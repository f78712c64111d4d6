// Mirror of the reference's tests/tree_traversal/tree_traversal.cu written against the
// drop-in header (include/grace/grace.h): GPU per-ray hit counts must equal the brute-force
// host loop over every (ray, sphere) pair, exactly.  Also runs the hitcounts dump
// (tests/hitcounts/hitcounts.cu) and a small projection through project_sph.
// Host generator and brute force come from the oracle (test infrastructure).
#include "grace/grace.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" {
void go_random_real4(uint32_t first, size_t n, const float* lo, const float* hi, void* out);
void go_brute_hitcounts(const void* rays, size_t n_rays, const void* s, size_t n, int* counts);
void go_brute_cumulative(const void* rays, size_t n_rays, const void* s, size_t n, float* out,
                         double* out64, int blocks);
}

int main(int argc, char* argv[])
{
    size_t N = 200000;
    size_t N_rays = 32 * 100;
    int max_per_leaf = 32;
    if (argc > 1) N = (size_t)std::strtol(argv[1], NULL, 10);
    if (argc > 2) N_rays = 32 * (size_t)std::strtol(argv[2], NULL, 10);
    if (argc > 3) max_per_leaf = (int)std::strtol(argv[3], NULL, 10);

    std::printf("Number of particles:     %zu\nNumber of rays:          %zu\n"
                "Max particles per leaf:  %d\n\n", N, N_rays, max_per_leaf);

    // tests/tree_traversal/tree_traversal.cu:51-54
    const grace::float4 low = grace::make_float4(-1E4f, -1E4f, -1E4f, 80.f);
    const grace::float4 high = grace::make_float4(1E4f, 1E4f, 1E4f, 400.f);
    std::vector<grace::float4> h_spheres(N);
    go_random_real4(0, N, &low.x, &high.x, h_spheres.data());

    grace::device_vector<grace::float4> d_spheres(h_spheres);
    grace::device_vector<grace::Ray> d_rays(N_rays);
    grace::device_vector<int> d_hit_counts(N_rays);
    grace::Tree d_tree(N, max_per_leaf);

    build_tree(d_spheres, low, high, d_tree);
    grace::detail::check(grace_rays_isotropic(N_rays, 0.f, 0.f, 0.f, 2E4f, 1234, d_rays.data(),
                                              nullptr));
    grace::trace_hitcounts_sph(d_rays, d_spheres, d_tree, d_hit_counts);

    h_spheres = d_spheres.to_host(); // sorted by the build
    std::vector<grace::Ray> h_rays = d_rays.to_host();
    std::vector<int> h_hit_counts = d_hit_counts.to_host();
    std::vector<int> ref(N_rays);
    go_brute_hitcounts(h_rays.data(), N_rays, h_spheres.data(), N, ref.data());

    double total = 0;
    size_t failed_rays = 0, failed_intersections = 0;
    for (size_t ri = 0; ri < N_rays; ++ri) {
        total += h_hit_counts[ri];
        if (ref[ri] != h_hit_counts[ri]) {
            ++failed_rays;
            failed_intersections += std::abs(ref[ri] - h_hit_counts[ri]);
        }
    }
    std::printf("Mean of %g hits per ray (device).\n\n", total / N_rays);

    // A wrong ray count must throw like the reference (bintree_trace.cuh:231-238).
    bool threw = false;
    try {
        grace::device_vector<grace::Ray> bad(33);
        grace::device_vector<int> out(33);
        grace::trace_hitcounts_sph(bad, d_spheres, d_tree, out);
    } catch (const std::invalid_argument&) { threw = true; }

    // project_sph on a unit-box snapshot: column densities against the brute-force sum.
    const size_t Np = 50000, side = 32;
    const grace::float4 plow = grace::make_float4(0.f, 0.f, 0.f, 0.01f);
    const grace::float4 phigh = grace::make_float4(1.f, 1.f, 1.f, 0.04f);
    std::vector<grace::float4> hp(Np);
    go_random_real4(0, Np, &plow.x, &phigh.x, hp.data());
    grace::device_vector<grace::float4> dp(hp);
    grace::device_vector<float> image;
    project_sph(dp, side, 32, image);
    hp = dp.to_host();
    grace::float4 mins, maxs;
    grace::min_max_vec4(dp, &mins, &maxs);
    mins.w = maxs.w = 0;
    grace::device_vector<grace::Ray> prays;
    orthogonal_rays_z(side, mins, maxs, prays);
    std::vector<grace::Ray> hr = prays.to_host();
    std::vector<float> ref32(hr.size());
    std::vector<double> ref64(hr.size());
    go_brute_cumulative(hr.data(), hr.size(), hp.data(), Np, ref32.data(), ref64.data(), 8);
    // Default evaluation (hardware sqrt, fp32 lerp): within the stated tolerance of the fp64 sum.
    std::vector<float> img = image.to_host();
    size_t image_mismatch = 0;
    for (size_t i = 0; i < img.size(); ++i)
        if (!(std::fabs(img[i] - ref64[i]) <= 3e-6 * std::fabs(ref64[i]))) ++image_mismatch;
    // The reference's per-hit arithmetic bit for bit: identical to the oracle's fp32 sum.
    grace_trace_set_exact_integrals(1);
    project_sph(dp, side, 32, image);
    grace_trace_set_exact_integrals(0);
    img = image.to_host();
    for (size_t i = 0; i < img.size(); ++i)
        if (img[i] != ref32[i]) ++image_mismatch;

    const bool ok = failed_rays == 0 && threw && image_mismatch == 0;
    if (ok) {
        std::printf("PASSED\n");
    } else {
        std::printf("FAILED\n%zu intersection tests failed over %zu rays; invalid_argument %s; "
                    "%zu of %zu projected pixels differ\n", failed_intersections, failed_rays,
                    threw ? "thrown" : "NOT thrown", image_mismatch, img.size());
    }
    return ok ? EXIT_SUCCESS : EXIT_FAILURE;
}
