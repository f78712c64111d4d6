// The ray generators of include/grace/cuda/gen_rays.cuh and the double4 key/sort overloads of
// build_sph.cuh through the drop-in header, checked on the host by their defining properties
// (the reference pins its generators statistically only: tests/isotropic_ray_stats).
#include "grace/grace.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { ++fails; std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); } } while (0)

int main()
{
    using namespace grace;
    // single octant: signs, unit length
    device_vector<Ray> d_rays(4096);
    uniform_random_rays_single_octant(d_rays, 1.f, 2.f, 3.f, 4.f, MPM, 77);
    std::vector<Ray> r = d_rays.to_host();
    size_t bad = 0;
    for (size_t i = 0; i < r.size(); ++i) {
        const float n = std::sqrt(r[i].dx * r[i].dx + r[i].dy * r[i].dy + r[i].dz * r[i].dz);
        bad += !(r[i].dx <= 0 && r[i].dy >= 0 && r[i].dz <= 0 && std::fabs(n - 1.f) < 1e-6f
                 && r[i].ox == 1.f && r[i].oy == 2.f && r[i].oz == 3.f && r[i].length == 4.f);
    }
    EXPECT(bad == 0);

    // one_to_many: every ray ends at its point; d_rays grows; bad sort type throws
    std::vector<double4> hp(1000);
    for (size_t i = 0; i < hp.size(); ++i) {
        hp[i].x = std::sin(0.37 * i) * 3; hp[i].y = std::cos(0.11 * i) * 2; hp[i].z = 0.001 * i; hp[i].w = 1;
    }
    device_vector<double4> d_pts(hp);
    device_vector<Ray> d_small(10);
    one_to_many_rays(d_small, 0.5f, 0.5f, -1.f, d_pts, NoSort);
    EXPECT(d_small.size() == hp.size());
    r = d_small.to_host();
    bad = 0;
    for (size_t i = 0; i < r.size(); ++i) {
        const double ex = r[i].ox + double(r[i].dx) * r[i].length - hp[i].x;
        const double ey = r[i].oy + double(r[i].dy) * r[i].length - hp[i].y;
        const double ez = r[i].oz + double(r[i].dz) * r[i].length - hp[i].z;
        bad += !(std::fabs(ex) < 1e-5 && std::fabs(ey) < 1e-5 && std::fabs(ez) < 1e-5);
    }
    EXPECT(bad == 0);
    one_to_many_rays(d_small, 0.5f, 0.5f, -1.f, d_pts, DirectionSort);
    one_to_many_rays(d_small, 0.5f, 0.5f, -1.f, d_pts, EndPointSort);
    bool threw = false;
    try { one_to_many_rays(d_small, 0.f, 0.f, 0.f, d_pts, RaySortType(7)); }
    catch (const std::invalid_argument&) { threw = true; }
    EXPECT(threw);

    // plane-parallel: the header comment's example
    device_vector<Ray> d_pp;
    plane_parallel_random_rays(d_pp, 20, 10, make_float3(5, 0, 10), make_float3(-5, 0, 0),
                               make_float3(0, 6, 0), 2.5f);
    r = d_pp.to_host();
    EXPECT(r.size() == 200);
    bad = 0;
    for (size_t i = 0; i < r.size(); ++i)
        bad += !(r[i].dx == 0 && r[i].dy == 0 && r[i].dz == -1.f && r[i].oz == 10.f
                 && r[i].ox >= 0 && r[i].ox <= 5 && r[i].oy >= 0 && r[i].oy <= 6);
    EXPECT(bad == 0);

    // orthographic projection == orthogonal_rays_z for the -z view of the unit box
    device_vector<Ray> d_o, d_z;
    orthographic_projection_rays(d_o, 16, 16, make_float3(.5f, .5f, 1.f), make_float3(.5f, .5f, 0.f),
                                 make_float3(0, 1, 0), 1.f, 2.f);
    orthogonal_rays_z(16, make_float4(0, 0, 0, 0), make_float4(1, 1, 1, 0), d_z);
    std::vector<Ray> a = d_o.to_host(), b = d_z.to_host();
    bad = 0;
    for (size_t i = 0; i < a.size(); ++i)
        bad += !(a[i].ox == b[i].ox && a[i].oy == b[i].oy && a[i].oz == b[i].oz && a[i].dz == b[i].dz
                 && a[i].length == b[i].length);
    EXPECT(bad == 0 && a.size() == 256);

    // pinhole: central direction along the view direction
    device_vector<Ray> d_p;
    pinhole_camera_rays(d_p, 33, 33, make_float3(0, 0, 5), make_float3(0, 0, 0), make_float3(0, 1, 0),
                        0.8f, 10.f);
    r = d_p.to_host();
    EXPECT(std::fabs(r[16 * 33 + 16].dz + 1.f) < 1e-6f);

    // double4 particles: 63-bit key sort leaves the keys of the sorted records non-decreasing
    device_vector<double4> d_s(hp);
    const float3 bot = make_float3(-3, -2, 0), top = make_float3(3, 2, 1);
    morton_keys63_sort_sph(d_s, bot, top);
    device_vector<uinteger64> d_k(hp.size());
    morton_keys_sph(d_s, bot, top, d_k);
    std::vector<uinteger64> k = d_k.to_host();
    bad = 0;
    for (size_t i = 1; i < k.size(); ++i) bad += k[i] < k[i - 1];
    EXPECT(bad == 0);

    // double4 build + trace through the mirror: tree over the sorted records, counts consistent
    // with the column densities (a ray has a positive sum iff it has hits)
    {
        std::vector<double4> hs(20000);
        for (size_t i = 0; i < hs.size(); ++i) {
            hs[i].x = 0.5 + 0.5 * std::sin(12.9898 * i); hs[i].y = 0.5 + 0.5 * std::sin(78.233 * i + 1.0);
            hs[i].z = 0.5 + 0.5 * std::sin(37.719 * i + 2.0); hs[i].w = 0.01 + 0.02 * (i % 7) / 7.0;
        }
        device_vector<double4> d_sph(hs);
        const float3 lo = make_float3(0, 0, 0), hi = make_float3(1, 1, 1);
        morton_keys30_sort_sph(d_sph, lo, hi);
        device_vector<float> d_del(hs.size() + 1);
        euclidean_deltas_sph(d_sph, d_del);
        Tree tree(hs.size(), 16);
        ALBVH_sph(d_sph, d_del, tree);
        device_vector<Ray> d_r(512);
        uniform_random_rays(d_r, 0.5f, 0.5f, 0.5f, 2.f, 5);
        device_vector<int> d_c(512);
        device_vector<double> d_s(512);
        trace_hitcounts_sph(d_r, d_sph, tree, d_c);
        trace_cumulative_sph(d_r, d_sph, tree, d_s);
        const std::vector<int> c = d_c.to_host();
        const std::vector<double> sm = d_s.to_host();
        bad = 0; long total = 0;
        for (size_t i = 0; i < c.size(); ++i) { bad += (c[i] > 0) != (sm[i] > 0.0); total += c[i]; }
        EXPECT(bad == 0 && total > 1000);
    }

    std::printf(fails ? "FAILED\n" : "PASSED\n");
    return fails ? EXIT_FAILURE : EXIT_SUCCESS;
}
