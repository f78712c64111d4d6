// The generic functor trace (include/grace/hip/trace.hpp) composed exactly like the
// reference's trace_hitcounts_sph / trace_cumulative_sph (include/grace/cuda/trace_sph.cuh:
// 58-110) plus a user-defined functor set the library has no built-in for ("largest sphere
// hit"), checked against the built-in C-ABI kernels and a host loop.  Build with hipcc.
#include "grace/hip/trace.hpp"

#include <cmath>
#include <cstdio>
#include <vector>

extern "C" void go_random_real4(uint32_t first, size_t n, const float* lo, const float* hi, void* out);
extern "C" const double* go_kernel_table(int* n);

struct RayData_maxr { int data; float rmax; };
struct OnHit_largest {     // user functor: remember the index of the largest sphere hit
    __device__ void operator()(int, const grace::Ray&, RayData_maxr& rd, int prim_idx,
                               const ::float4& s, int, const grace::gpu::BoundIter<char>) const
    { if (s.w > rd.rmax) { rd.rmax = s.w; rd.data = prim_idx; } }
};
struct RayEntry_maxr {
    __device__ void operator()(int, const grace::Ray&, RayData_maxr& rd, const grace::gpu::BoundIter<char>) const
    { rd.data = -1; rd.rmax = 0.f; }
};

int main()
{
    const size_t N = 40000, R = 32 * 37; // not a multiple of 64
    const grace::float4 lo = grace::make_float4(0, 0, 0, 0), hi = grace::make_float4(1, 1, 1, 0.06f);
    std::vector<grace::float4> hs(N);
    go_random_real4(0, N, &lo.x, &hi.x, hs.data());
    grace::device_vector<grace::float4> d_spheres(hs);
    grace::Tree tree(N, 16);
    build_tree(d_spheres, lo, hi, tree);
    grace::device_vector<grace::Ray> d_rays(R);
    grace::detail::check(grace_rays_isotropic(R, 0.5f, 0.5f, 0.5f, 2.f, 99, d_rays.data(), nullptr));
    const ::float4* prims = reinterpret_cast<const ::float4*>(d_spheres.data());

    // hit counts through functors vs the built-in kernel
    grace::device_vector<int> c_generic(R), c_builtin(R);
    grace::trace<grace::RayData_datum<int>>(d_rays.data(), R, prims, N, tree, 0, grace::Init_null(),
        grace::Intersect_sphere_bool(), grace::OnHit_increment(), grace::RayEntry_null(),
        grace::RayExit_to_array<int>(c_generic.data()));
    grace::trace_hitcounts_sph(d_rays, d_spheres, tree, c_builtin);
    const std::vector<int> a = c_generic.to_host(), b = c_builtin.to_host();
    size_t bad_counts = 0; long total = 0;
    for (size_t i = 0; i < R; ++i) { bad_counts += a[i] != b[i]; total += a[i]; }

    // cumulative through functors (table in user LDS) vs the built-in kernel, bit for bit
    int n_table = 0;
    const double* table = go_kernel_table(&n_table);
    grace::device_vector<double> d_table(std::vector<double>(table, table + n_table));
    grace::device_vector<float> s_generic(R), s_builtin(R);
    grace::trace<grace::RayData_sphere<float, float>>(d_rays.data(), R, prims, N, tree,
        sizeof(double) * n_table, grace::InitGlobalToSmem<double>(d_table.data(), n_table),
        grace::Intersect_sphere_b2dist(), grace::OnHit_sphere_cumulate(n_table),
        grace::RayEntry_null(), grace::RayExit_to_array<float>(s_generic.data()));
    grace::trace_cumulative_sph(d_rays, d_spheres, tree, s_builtin);
    const std::vector<float> sa = s_generic.to_host(), sb = s_builtin.to_host();
    // The generic kernel keeps the reference's single running sum; the built-in kernel states the
    // block-ordered sum: equal to a few ulp.
    size_t bad_sums = 0;
    for (size_t i = 0; i < R; ++i) bad_sums += !(std::fabs(sa[i] - sb[i]) <= 2e-6f * std::fabs(sb[i]));

    // user-defined functor vs a host loop
    grace::device_vector<int> d_big(R);
    grace::trace<RayData_maxr>(d_rays.data(), R, prims, N, tree, 0, grace::Init_null(),
        grace::Intersect_sphere_bool(), OnHit_largest(), RayEntry_maxr(),
        grace::RayExit_to_array<int>(d_big.data()));
    const std::vector<int> big = d_big.to_host();
    const std::vector<grace::Ray> hr = d_rays.to_host();
    hs = d_spheres.to_host();
    size_t bad_big = 0;
    for (size_t r = 0; r < R; ++r) {
        int best = -1; float rmax = 0.f;
        const grace::Ray& ray = hr[r];
        for (size_t i = 0; i < N; ++i) {
            const grace::float4& s = hs[i];
            const float px = s.x - ray.ox, py = s.y - ray.oy, pz = s.z - ray.oz;
            const float dp = px * ray.dx + py * ray.dy + pz * ray.dz;
            const float bx = px - dp * ray.dx, by = py - dp * ray.dy, bz = pz - dp * ray.dz;
            const float b2 = bx * bx + by * by + bz * bz;
            if (!(b2 >= s.w * s.w) && !(dp < 0.f) && !(dp >= ray.length) && s.w > rmax) { rmax = s.w; best = int(i); }
        }
        bad_big += best != big[r];
    }
    std::printf("rays %zu hits %ld: count mismatches %zu, sum mismatches %zu, user-functor mismatches %zu\n",
                R, total, bad_counts, bad_sums, bad_big);
    const bool ok = bad_counts == 0 && bad_sums == 0 && bad_big == 0 && total > 0;
    std::printf(ok ? "PASSED\n" : "FAILED\n");
    return ok ? 0 : 1;
}
