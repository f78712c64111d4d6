// The generic forms of the drop-in headers against the sphere-specialised ones and host loops
// (authored here; run by tests/test_gpu_dropin.py, compiled on the CPU by tests/test_capi_cpu.py):
//   * grace::min_max_x/y/z/w, min_vec2/3/4, max_vec2/3/4 over float4 / float3 / double4 / int4 /
//     a caller's struct, in their device_vector, pointer, device-iterator and host_vector forms
//     (reference include/grace/cuda/util/extrema.cuh:190-772; tests/profile_trace_gadget/
//     profile_trace_gadget.cu:83 calls min_max_x);
//   * grace::morton_keys(prims, bot, top, keys, CentroidFunc) and its bounds-free form with the
//     stock CentroidSphere, with PrimitiveCentroid<float4, AABBSphere> and with a caller's functor,
//     against grace::morton_key on the host (kernels/morton.cuh:97-189, generic/morton.h);
//   * grace::compute_deltas with DeltaXOR / DeltaEuclidean / DeltaSurfaceArea against
//     XOR_deltas_sph / euclidean_deltas_sph / surface_area_deltas_sph (albvh.cuh:949-978);
//   * grace::build_ALBVH(tree, prims, deltas, AABBFunc) with AABBSphere, with a caller's functor
//     doing the same arithmetic (the generic box path), with thrust::greater on negated deltas --
//     all the tree ALBVH_sph builds (albvh.cuh:986-1072);
//   * weighted_exclusive_segmented_scan<double> (scan.cuh:43-58).
//   dropin_generic <n_spheres>
#include "grace/cuda/build_sph.cuh"
#include "grace/cuda/kernels/albvh.cuh"
#include "grace/cuda/kernels/morton.cuh"
#include "grace/cuda/nodes.h"
#include "grace/cuda/scan.cuh"
#include "grace/cuda/util/extrema.cuh"
#include "grace/generic/functors/aabb.h"
#include "grace/generic/functors/albvh.h"
#include "grace/generic/functors/centroid.h"
#include "grace/generic/morton.h"

#include "helper/random.cuh"

#include <thrust/device_vector.h>
#include <thrust/host_vector.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static int failures = 0;
#define CHECK(cond, what) do { if (!(cond)) { std::printf("FAILED %s (line %d)\n", what, __LINE__); ++failures; } } while (0)

// A caller's own functors: the sphere's box and centre, spelled differently from the stock ones.
struct MyBox
{
    __host__ __device__ void operator()(const float4 s, float3* bot, float3* top) const
    {
        *bot = make_float3(s.x - s.w, s.y - s.w, s.z - s.w);
        *top = make_float3(s.x + s.w, s.y + s.w, s.z + s.w);
    }
};
struct MyCentre
{
    __host__ __device__ float3 operator()(const float4 s) const { return make_float3(s.x, s.y, s.z); }
};
struct Particle { float x, y; int id; };   // a caller's record with two leading float components

template <typename T>
static bool same_bits(const thrust::device_vector<T>& a, const thrust::device_vector<T>& b)
{
    thrust::host_vector<T> ha = a, hb = b;
    return ha.size() == hb.size() && std::memcmp(ha.data(), hb.data(), ha.size() * sizeof(T)) == 0;
}

static bool same_tree(const grace::Tree& a, const grace::Tree& b)
{
    int ra = -1, rb = -2;
    GRACE_HIP_CHECK(hipMemcpy(&ra, a.root_index_ptr, sizeof(int), hipMemcpyDeviceToHost));
    GRACE_HIP_CHECK(hipMemcpy(&rb, b.root_index_ptr, sizeof(int), hipMemcpyDeviceToHost));
    return ra == rb && same_bits(a.nodes, b.nodes) && same_bits(a.leaves, b.leaves);
}

int main(int argc, char* argv[])
{
    const size_t N = argc > 1 ? size_t(std::atol(argv[1])) : 50000;

    thrust::device_vector<float4> d_spheres;
    random_real4(make_float4(-3.f, 0.f, 10.f, 0.01f), make_float4(5.f, 1.f, 12.f, 0.2f), N, d_spheres);
    thrust::host_vector<float4> h_spheres = d_spheres;

    // ---- extrema ----------------------------------------------------------------------------
    {
        float lo[4] = { 1e30f, 1e30f, 1e30f, 1e30f }, hi[4] = { -1e30f, -1e30f, -1e30f, -1e30f };
        for (size_t i = 0; i < N; ++i) {
            const float c[4] = { h_spheres[i].x, h_spheres[i].y, h_spheres[i].z, h_spheres[i].w };
            for (int k = 0; k < 4; ++k) { lo[k] = c[k] < lo[k] ? c[k] : lo[k]; hi[k] = c[k] > hi[k] ? c[k] : hi[k]; }
        }
        float a, b; double da, db;
        grace::min_max_x(d_spheres, &a, &b); CHECK(a == lo[0] && b == hi[0], "min_max_x(device_vector)");
        grace::min_max_y(d_spheres, &da, &db); CHECK(da == lo[1] && db == hi[1], "min_max_y -> double");
        grace::min_max_z(thrust::raw_pointer_cast(d_spheres.data()), N, &a, &b); CHECK(a == lo[2] && b == hi[2], "min_max_z(pointer)");
        const float4* cp = thrust::raw_pointer_cast(d_spheres.data());
        grace::min_max_w(cp, N, &a, &b); CHECK(a == lo[3] && b == hi[3], "min_max_w(const pointer)");
        grace::min_max_x(d_spheres.begin(), N, &a, &b); CHECK(a == lo[0] && b == hi[0], "min_max_x(device iterator)");
        grace::min_max_w(h_spheres, &a, &b); CHECK(a == lo[3] && b == hi[3], "min_max_w(host_vector)");
        float2 m2, M2; float3 m3, M3; float4 m4, M4; double4 M4d;
        grace::min_vec2(d_spheres, &m2); grace::max_vec2(d_spheres, &M2);
        CHECK(m2.x == lo[0] && m2.y == lo[1] && M2.x == hi[0] && M2.y == hi[1], "min/max_vec2");
        grace::min_vec3(d_spheres, &m3); grace::max_vec3(cp, N, &M3);
        CHECK(m3.x == lo[0] && m3.z == lo[2] && M3.y == hi[1] && M3.z == hi[2], "min/max_vec3");
        grace::min_vec4(d_spheres, &m4); grace::max_vec4(d_spheres, &M4d); grace::max_vec4(h_spheres, &M4);
        CHECK(m4.x == lo[0] && m4.w == lo[3] && M4d.w == hi[3] && M4.w == hi[3] && M4.y == hi[1], "min/max_vec4");

        thrust::host_vector<double4> h_d(N); thrust::host_vector<int4> h_i(N); thrust::host_vector<Particle> h_p(N);
        double dlo = 1e300, dhi = -1e300; int ilo = 0x7fffffff, ihi = -0x7fffffff; float plo = 1e30f, phi = -1e30f;
        for (size_t i = 0; i < N; ++i) {
            h_d[i] = make_double4(h_spheres[i].x * 1.0000001, h_spheres[i].y, h_spheres[i].z * -3.0, 1.0);
            h_i[i] = make_int4(int(i) - 7, int((i * 2654435761u) & 0xffff) - 30000, 3, -int(i % 1001));
            h_p[i].x = h_spheres[i].w; h_p[i].y = -h_spheres[i].x; h_p[i].id = int(i);
            dlo = h_d[i].z < dlo ? h_d[i].z : dlo; dhi = h_d[i].z > dhi ? h_d[i].z : dhi;
            ilo = h_i[i].y < ilo ? h_i[i].y : ilo; ihi = h_i[i].y > ihi ? h_i[i].y : ihi;
            plo = h_p[i].y < plo ? h_p[i].y : plo; phi = h_p[i].y > phi ? h_p[i].y : phi;
        }
        thrust::device_vector<double4> d_d = h_d; thrust::device_vector<int4> d_i = h_i; thrust::device_vector<Particle> d_p = h_p;
        double x0, x1; int i0, i1;
        grace::min_max_z(d_d, &x0, &x1); CHECK(x0 == dlo && x1 == dhi, "min_max_z(double4)");
        grace::min_max_y(d_i, &i0, &i1); CHECK(i0 == ilo && i1 == ihi, "min_max_y(int4)");
        grace::min_max_y(d_p, &a, &b); CHECK(a == plo && b == phi, "min_max_y(caller's struct)");
        int4 im; grace::min_vec4(d_i, &im); CHECK(im.x == -7 && im.z == 3 && im.w == -1000, "min_vec4(int4)");
    }

    // ---- Morton keys ------------------------------------------------------------------------
    const float3 bot = make_float3(-3.f, 0.f, 10.f), top = make_float3(5.f, 1.f, 12.f);
    {
        thrust::device_vector<grace::uinteger32> k_sph(N), k_stock(N), k_box(N), k_mine(N), k_free(N), k_free2(N);
        thrust::device_vector<grace::uinteger64> k63_sph(N), k63_mine(N);
        grace::morton_keys_sph(d_spheres, bot, top, k_sph);
        grace::morton_keys(d_spheres, bot, top, k_stock, grace::CentroidSphere());
        grace::morton_keys(d_spheres, bot, top, k_mine, MyCentre());
        grace::morton_keys(thrust::raw_pointer_cast(d_spheres.data()), N, bot, top, k_box.begin(),
                           grace::PrimitiveCentroid<float4, grace::AABBSphere>());
        CHECK(same_bits(k_sph, k_stock), "morton_keys(CentroidSphere) == morton_keys_sph");
        CHECK(same_bits(k_sph, k_mine), "morton_keys(caller's centroid) == morton_keys_sph");
        // host keys with the same scale arithmetic (kernels/morton.cuh:43-50,104-113)
        thrust::host_vector<grace::uinteger32> hk = k_sph, hb = k_box;
        const float sx = 1023 / (top.x - bot.x), sy = 1023 / (top.y - bot.y), sz = 1023 / (top.z - bot.z);
        bool ok = true, okb = true;
        for (size_t i = 0; i < N; ++i) {
            const float4 s = h_spheres[i];
            ok = ok && hk[i] == grace::morton_key(grace::uinteger32(sx * (s.x - bot.x)), grace::uinteger32(sy * (s.y - bot.y)),
                                                  grace::uinteger32(sz * (s.z - bot.z)));
            float3 b3, t3; grace::AABBSphere()(s, &b3, &t3);
            const float3 c = grace::detail::AABB_centroid(b3, t3);
            okb = okb && hb[i] == grace::morton_key(grace::uinteger32(sx * (c.x - bot.x)), grace::uinteger32(sy * (c.y - bot.y)),
                                                    grace::uinteger32(sz * (c.z - bot.z)));
        }
        CHECK(ok, "device keys == host grace::morton_key");
        CHECK(okb, "PrimitiveCentroid<float4, AABBSphere> keys == host");
        float3 fb, ft, fb2, ft2;
        grace::morton_keys_sph(d_spheres, k_free);
        grace::morton_keys(d_spheres, k_free2, grace::CentroidSphere(), &fb, &ft);
        CHECK(same_bits(k_free, k_free2), "bounds-free morton_keys == morton_keys_sph");
        grace::min_vec3(d_spheres, &fb2); grace::max_vec3(d_spheres, &ft2);
        CHECK(fb.x == fb2.x && fb.y == fb2.y && fb.z == fb2.z && ft.x == ft2.x && ft.z == ft2.z, "returned centroid bounds");
        const double3 dbot = make_double3(-3., 0., 10.), dtop = make_double3(5., 1., 12.);
        grace::morton_keys_sph(d_spheres, dbot, dtop, k63_sph);
        grace::morton_keys(d_spheres, dbot, dtop, k63_mine, MyCentre());
        CHECK(same_bits(k63_sph, k63_mine), "63-bit keys, double3 bounds, caller's centroid");
    }

    // ---- sort, deltas, trees ---------------------------------------------------------------
    grace::morton_keys30_sort_sph(d_spheres, bot, top);
    thrust::device_vector<grace::uinteger32> d_keys(N);
    grace::morton_keys_sph(d_spheres, bot, top, d_keys);
    {
        thrust::device_vector<float> e_sph(N + 1), e_gen(N + 1), a_sph(N + 1), a_gen(N + 1);
        thrust::device_vector<grace::uinteger32> x_sph(N + 1), x_gen(N + 1);
        grace::euclidean_deltas_sph(d_spheres, e_sph);
        grace::compute_deltas(d_spheres, e_gen, grace::DeltaEuclidean<const float4*, grace::CentroidSphere>());
        CHECK(same_bits(e_sph, e_gen), "compute_deltas(DeltaEuclidean) == euclidean_deltas_sph");
        grace::surface_area_deltas_sph(d_spheres, a_sph);
        grace::compute_deltas(d_spheres, a_gen, grace::DeltaSurfaceArea<const float4*, grace::AABBSphere>());
        CHECK(same_bits(a_sph, a_gen), "compute_deltas(DeltaSurfaceArea) == surface_area_deltas_sph");
        grace::XOR_deltas_sph(d_keys, x_sph);
        grace::compute_deltas(d_keys, x_gen, grace::DeltaXOR());
        CHECK(same_bits(x_sph, x_gen), "compute_deltas(DeltaXOR) == XOR_deltas_sph");

        for (int mpl = 1; mpl <= 32; mpl *= 8) {
            grace::Tree t_sph(N, mpl), t_stock(N, mpl), t_mine(N, mpl), t_gt(N, mpl), t_xor(N, mpl), t_xor_mine(N, mpl);
            grace::ALBVH_sph(d_spheres, e_sph, t_sph);
            grace::build_ALBVH(t_stock, d_spheres, e_sph, grace::AABBSphere());
            grace::build_ALBVH(t_mine, thrust::raw_pointer_cast(d_spheres.data()), e_sph.begin(), MyBox(), true);
            CHECK(same_tree(t_sph, t_stock), "build_ALBVH(AABBSphere) == ALBVH_sph");
            CHECK(same_tree(t_sph, t_mine), "build_ALBVH(caller's AABB functor) == ALBVH_sph");
            thrust::host_vector<float> h_e = e_sph;
            for (size_t i = 0; i < h_e.size(); ++i) h_e[i] = -h_e[i];
            thrust::device_vector<float> e_neg = h_e;
            grace::build_ALBVH(t_gt, d_spheres, e_neg, thrust::greater<float>(), MyBox());
            CHECK(same_tree(t_sph, t_gt), "build_ALBVH(thrust::greater, negated deltas) == ALBVH_sph");
            grace::ALBVH_sph(d_spheres, x_sph, t_xor);
            grace::build_ALBVH(t_xor_mine, d_spheres, x_sph, thrust::less<grace::uinteger32>(), MyBox());
            CHECK(same_tree(t_xor, t_xor_mine), "build_ALBVH(XOR deltas, caller's functor) == ALBVH_sph");
        }
        bool threw = false;
        try { grace::Tree t_bad(8, 8); thrust::device_vector<float4> few(d_spheres.begin(), d_spheres.begin() + 8);
              thrust::device_vector<float> dl(9, 1.f); grace::build_ALBVH(t_bad, few, dl, MyBox()); }
        catch (const std::invalid_argument&) { threw = true; }
        CHECK(threw, "build_ALBVH throws std::invalid_argument when N <= max_per_leaf");
    }

    // ---- weighted segmented scan in double -----------------------------------------------------
    {
        const size_t n_seg = 257, n = 40000;
        thrust::host_vector<int> h_off(n_seg);
        for (size_t s = 0; s < n_seg; ++s) h_off[s] = int((s * s * n) / (n_seg * n_seg));
        thrust::host_vector<double> h_x(n), h_w(16), h_ref(n);
        thrust::host_vector<unsigned int> h_map(n);
        for (size_t k = 0; k < 16; ++k) h_w[k] = double(k + 1);
        for (size_t i = 0; i < n; ++i) { h_x[i] = double(1 + i % 9); h_map[i] = unsigned((i * 7) % 16); }
        for (size_t s = 0; s < n_seg; ++s) {
            const size_t b = h_off[s], e = s + 1 < n_seg ? size_t(h_off[s + 1]) : n;
            double run = 0.0;
            for (size_t i = b; i < e; ++i) { h_ref[i] = run; run += h_x[i] * h_w[h_map[i]]; }
        }
        thrust::device_vector<double> d_x = h_x, d_w = h_w, d_sum(n);
        thrust::device_vector<unsigned int> d_map = h_map;
        thrust::device_vector<int> d_off = h_off;
        grace::weighted_exclusive_segmented_scan(d_x, d_w, d_map, d_off, d_sum);
        thrust::host_vector<double> h_sum = d_sum;
        bool ok = true;
        for (size_t i = 0; i < n; ++i) ok = ok && h_sum[i] == h_ref[i];
        CHECK(ok, "weighted_exclusive_segmented_scan<double> == host loop (integer data)");
        thrust::host_vector<float> h_xf(n), h_wf(16);
        for (size_t i = 0; i < n; ++i) h_xf[i] = float(h_x[i]);
        for (size_t k = 0; k < 16; ++k) h_wf[k] = float(h_w[k]);
        thrust::device_vector<float> d_xf = h_xf, d_wf = h_wf, d_sumf(n);
        grace::weighted_exclusive_segmented_scan(d_xf, d_wf, d_map, d_off, d_sumf);
        thrust::host_vector<float> h_sumf = d_sumf;
        ok = true;
        for (size_t i = 0; i < n; ++i) ok = ok && double(h_sumf[i]) == h_ref[i];
        CHECK(ok, "weighted_exclusive_segmented_scan<float> == host loop (integer data)");
    }

    std::printf(failures ? "%d check(s) FAILED\n" : "PASSED\n", failures);
    return failures ? EXIT_FAILURE : EXIT_SUCCESS;
}
