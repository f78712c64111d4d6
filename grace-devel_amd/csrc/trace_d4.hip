// Traversal for double4 spheres (Real4 = double4, Real = double): the reference's
// trace_hitcounts_sph / trace_cumulative_sph templates instantiated in double
// (include/grace/cuda/trace_sph.cuh:57-110): sphere_hit<double4, double>
// (generic/intersect.h:9-55, ray members are float, everything else double),
// OnHit_sphere_cumulate with Real = double (functors/trace.cuh:164-186: ir = 1.f / w,
// b = (N - 1) * (sqrt(b2) * ir), lerp<double> with the device branch's fma, integral *= ir * ir),
// one running double sum per ray in ascending primitive index (RayData_sphere<double, double>).
//
// A compact, straightforward kernel -- coverage of the double-precision instantiation, not the
// tuned path of trace.hip: one 64-lane wavefront per 64 consecutive rays (caller order), the
// packet's stack in LDS, the reference's per-ray slab test (float boxes), and every sphere of
// every entered leaf tested by all 64 rays (wave-uniform 32-byte sphere loads).  fp64 vector
// work throughout; no MFMA.
#include "common.hpp"

using namespace grace_hip;

namespace {

constexpr int D4_BLOCK = 256;
constexpr int D4_TABLE = 51;

// include/grace/cuda/trace_sph.cuh:32-48 (kernel_integrals.h)
__constant__ double c_table_d4[D4_TABLE] = {
    1.90986019771937, 1.90563449910964, 1.89304415940934, 1.87230928086763,
    1.84374947679902, 1.80776276033034, 1.76481079856299, 1.71540816859939,
    1.66011373131439, 1.59952322363667, 1.53426266082279, 1.46498233888091,
    1.39235130929287, 1.31705223652377, 1.23977618317103, 1.16121278415369,
    1.08201943664419, 1.00288866679720, 0.924475767210246, 0.847415371038733,
    0.772316688105931, 0.699736940377312, 0.630211918937167, 0.564194562399538,
    0.502076205853037, 0.444144023534733, 0.390518196140658, 0.341148855945766,
    0.295941946237307, 0.254782896476983, 0.217538645099225, 0.184059547649710,
    0.154181189781890, 0.127726122453554, 0.104505535066266,
    8.432088120445191E-002, 6.696547102921641E-002, 5.222604427168923E-002,
    3.988433820097490E-002, 2.971866601747601E-002, 2.150552303075515E-002,
    1.502124104014533E-002, 1.004371608622562E-002, 6.354242122978656E-003,
    3.739494884706115E-003, 1.993729589156428E-003, 9.212900163813992E-004,
    3.395908945333921E-004, 8.287326418242995E-005, 7.387919939044624E-006,
    0.000000000000000E+000
};

__device__ __forceinline__ int imin_(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax_(int a, int b) { return a > b ? a : b; }

// AABBs_hit, include/grace/cuda/device/intersect.cuh:10-40 (integer min/max of the float bits,
// device/intrinsics.cuh:8-51) -- as in trace.hip.
__device__ __forceinline__ void aabbs_hit_d4(const float ix, const float iy, const float iz,
                                             const float ox, const float oy, const float oz,
                                             const float len, const float4 L, const float4 R,
                                             const float4 Z, bool& hit_l, bool& hit_r)
{
    const float bx_L = (L.x - ox) * ix, tx_L = (L.y - ox) * ix;
    const float by_L = (L.z - oy) * iy, ty_L = (L.w - oy) * iy;
    const float bz_L = (Z.x - oz) * iz, tz_L = (Z.y - oz) * iz;
    const float bx_R = (R.x - ox) * ix, tx_R = (R.y - ox) * ix;
    const float by_R = (R.z - oy) * iy, ty_R = (R.w - oy) * iy;
    const float bz_R = (Z.z - oz) * iz, tz_R = (Z.w - oz) * iz;
    const int zero = __float_as_int(0.0f), ilen = __float_as_int(len);
    const int tmin_L = imax_(imax_(__float_as_int(fminf(bx_L, tx_L)), __float_as_int(fminf(by_L, ty_L))),
                             imax_(imin_(__float_as_int(bz_L), __float_as_int(tz_L)), zero));
    const int tmax_L = imin_(imin_(__float_as_int(fmaxf(bx_L, tx_L)), __float_as_int(fmaxf(by_L, ty_L))),
                             imin_(imax_(__float_as_int(bz_L), __float_as_int(tz_L)), ilen));
    const int tmin_R = imax_(imax_(__float_as_int(fminf(bx_R, tx_R)), __float_as_int(fminf(by_R, ty_R))),
                             imax_(imin_(__float_as_int(bz_R), __float_as_int(tz_R)), zero));
    const int tmax_R = imin_(imin_(__float_as_int(fmaxf(bx_R, tx_R)), __float_as_int(fmaxf(by_R, ty_R))),
                             imin_(imax_(__float_as_int(bz_R), __float_as_int(tz_R)), ilen));
    hit_r = __int_as_float(tmax_R) >= __int_as_float(tmin_R);
    hit_l = __int_as_float(tmax_L) >= __int_as_float(tmin_L);
}

template <bool CUMULATIVE>
__global__ __launch_bounds__(D4_BLOCK) void trace_d4_kernel(const float* __restrict__ rays, int n_rays,
                                                            const double* __restrict__ spheres,
                                                            const float4* __restrict__ nodes,
                                                            const int4* __restrict__ leaves,
                                                            const int* __restrict__ root, int n_nodes,
                                                            int* __restrict__ out_counts,
                                                            double* __restrict__ out_sums,
                                                            int* __restrict__ status)
{
    __shared__ double s_table[D4_TABLE];
    __shared__ int s_stack[D4_BLOCK / 64][128];
    if (CUMULATIVE) {
        if (threadIdx.x < D4_TABLE) s_table[threadIdx.x] = c_table_d4[threadIdx.x];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int first_ray = (blockIdx.x * (D4_BLOCK / 64) + wave) * 64;
    if (first_ray >= n_rays) return;
    const bool valid = first_ray + lane < n_rays;
    const int ray_index = valid ? first_ray + lane : n_rays - 1;   // tail lanes re-trace the last ray
    const float* rp = rays + 7 * size_t(ray_index);
    const float dx = rp[0], dy = rp[1], dz = rp[2], ox = rp[3], oy = rp[4], oz = rp[5], len = rp[6];
    const float ix = 1.f / dx, iy = 1.f / dy, iz = 1.f / dz;       // bintree_trace.cuh:111-114
    const double rx = dx, ry = dy, rz = dz;

    int count = 0;
    double sum = 0.0;
    int* stack = s_stack[wave];
    int sp = 0;
    bool overflow = false;
    stack[0] = *root;
    while (sp >= 0) {
        const int idx = __builtin_amdgcn_readfirstlane(stack[sp]);
        --sp;
        if (idx < n_nodes) {
            const float4* np = nodes + 4 * size_t(idx);
            const float4 n0 = np[0], L = np[1], R = np[2], Z = np[3];
            bool hit_l, hit_r;
            aabbs_hit_d4(ix, iy, iz, ox, oy, oz, len, L, R, Z, hit_l, hit_r);
            // right first, so that the left subtree -- lower primitive indices -- is walked first
            if (__builtin_amdgcn_ballot_w64(hit_r) != 0ull) {
                if (sp < 127) { ++sp; if (lane == 0) stack[sp] = __float_as_int(n0.y); } else overflow = true;
            }
            if (__builtin_amdgcn_ballot_w64(hit_l) != 0ull) {
                if (sp < 127) { ++sp; if (lane == 0) stack[sp] = __float_as_int(n0.x); } else overflow = true;
            }
            __builtin_amdgcn_wave_barrier();
        } else {
            const int4 leaf = leaves[idx - n_nodes];
            for (int i = 0; i < leaf.y; ++i) {
                const double* s = spheres + 4 * size_t(leaf.x + i);
                const double sx = s[0], sy = s[1], sz = s[2], sw = s[3];
                // sphere_hit<double4, double>, generic/intersect.h:16-54
                const double px = sx - ox, py = sy - oy, pz = sz - oz;
                const double dot_p = px * rx + py * ry + pz * rz;
                const double bx = px - dot_p * rx, by = py - dot_p * ry, bz = pz - dot_p * rz;
                const double b2 = bx * bx + by * by + bz * bz;
                const bool hit = !(b2 >= sw * sw) && !(dot_p < 0.0f) && !(dot_p >= len);
                if (!CUMULATIVE) {
                    count += hit ? 1 : 0;
                } else if (hit) {
                    // OnHit_sphere_cumulate, Real = double; lerp<double>, device branch
                    const double ir = 1.f / sw;
                    double x = (D4_TABLE - 1) * (sqrt(b2) * ir);
                    int x_idx = static_cast<int>(x);
                    if (x_idx >= D4_TABLE - 1) { x = double(D4_TABLE - 1); x_idx = D4_TABLE - 2; }
                    const double y0 = s_table[x_idx], y1 = s_table[x_idx + 1];
                    const double t = x - x_idx;
                    double integral = __builtin_fma(t, y1 - y0, y0);
                    integral *= (ir * ir);
                    sum += integral;
                }
            }
        }
    }
    if (overflow && lane == 0) *status = GRACE_STACK_OVERFLOW;
    if (!valid) return;
    if (CUMULATIVE) out_sums[ray_index] = sum;
    else out_counts[ray_index] = count;
}

int* g_status_d4 = nullptr;

template <bool CUMULATIVE>
grace_status launch_d4(const void* d_rays, size_t n_rays, const double* d_spheres, size_t n_spheres,
                       const int* d_nodes, size_t n_nodes, const int* d_leaves, const int* d_root,
                       int* d_counts, double* d_sums, hipStream_t stream)
{
    GRACE_REQUIRE(d_rays && d_spheres && d_nodes && d_leaves && d_root, "trace (double4): null pointer");
    GRACE_REQUIRE(CUMULATIVE ? d_sums != nullptr : d_counts != nullptr, "trace (double4): null output");
    GRACE_REQUIRE(n_rays > 0 && n_rays < (size_t(1) << 31), "trace (double4): bad ray count");
    GRACE_REQUIRE(n_nodes >= 1 && n_nodes < (size_t(1) << 30) && n_spheres > 0, "trace (double4): bad tree");
    if (!g_status_d4) {
        GRACE_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&g_status_d4), sizeof(int)));
        GRACE_TRY_HIP(hipMemsetAsync(g_status_d4, 0, sizeof(int), stream));
    }
    const int packets = int((n_rays + 63) / 64);
    trace_d4_kernel<CUMULATIVE><<<(packets + D4_BLOCK / 64 - 1) / (D4_BLOCK / 64), D4_BLOCK, 0, stream>>>(
        static_cast<const float*>(d_rays), int(n_rays), d_spheres,
        reinterpret_cast<const float4*>(d_nodes), reinterpret_cast<const int4*>(d_leaves), d_root,
        int(n_nodes), d_counts, d_sums, g_status_d4);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

} // namespace

extern "C" {

grace_status grace_trace_hitcounts_d4(const void* d_rays, size_t n_rays, const double* d_spheres,
                                      size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                      const int* d_leaves, const int* d_root, int* d_hit_counts,
                                      grace_stream stream)
{
    return launch_d4<false>(d_rays, n_rays, d_spheres, n_spheres, d_nodes, n_nodes, d_leaves, d_root,
                            d_hit_counts, nullptr, as_stream(stream));
}

grace_status grace_trace_cumulative_d4(const void* d_rays, size_t n_rays, const double* d_spheres,
                                       size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                       const int* d_leaves, const int* d_root, double* d_sums,
                                       grace_stream stream)
{
    return launch_d4<true>(d_rays, n_rays, d_spheres, n_spheres, d_nodes, n_nodes, d_leaves, d_root,
                           nullptr, d_sums, as_stream(stream));
}

grace_status grace_trace_status_d4(grace_stream stream)
{
    if (!g_status_d4) return GRACE_OK;
    int h = 0;
    GRACE_TRY_HIP(hipMemcpyAsync(&h, g_status_d4, sizeof(int), hipMemcpyDeviceToHost, as_stream(stream)));
    GRACE_TRY_HIP(hipStreamSynchronize(as_stream(stream)));
    if (h != 0) {
        GRACE_TRY_HIP(hipMemsetAsync(g_status_d4, 0, sizeof(int), as_stream(stream)));
        return set_error(GRACE_STACK_OVERFLOW, __FILE__, __LINE__, "traversal stack exhausted (double4 trace)");
    }
    return GRACE_OK;
}

} // extern "C"
