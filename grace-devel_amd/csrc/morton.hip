// Morton keys and centroid bounds for gfx950.
//
// Replaces morton_keys_kernel (reference include/grace/cuda/kernels/morton.cuh:30-55), the
// host scale computation (morton.cuh:97-119) and, for the bounds-free overloads, the
// compute_centroids kernel + two thrust::reduce passes (morton.cuh:139-174,
// kernels/aabb.cuh:14-48) -- here one fused streaming pass.
//
// HBM-bound streaming work: one float4 (16 B) coalesced load per lane, one 4/8 B store.
// Algorithmic bytes: 20 B/sphere (30-bit keys), 24 B/sphere (63-bit); bounds pass 16 B/sphere.
#include "common.hpp"

#include "grace/generic/morton.h"

using namespace grace_hip;

namespace {

// The key arithmetic itself is the product header's (include/grace/generic/{bits,morton}.h:
// host- and device-callable, pinned by the reference's known-answer tests).
template <typename Key>
struct Interleave;

template <>
struct Interleave<uint32_t> {
    static __device__ __forceinline__ uint32_t key(uint32_t x, uint32_t y, uint32_t z)
    {
        return grace::morton_key(x, y, z);
    }
};

template <>
struct Interleave<uint64_t> {
    static __device__ __forceinline__ uint64_t key(uint64_t x, uint64_t y, uint64_t z)
    {
        return grace::morton_key(x, y, z);
    }
};

// Real is the precision of the bounds (Real3 in the reference): scale * (centre - min) is
// evaluated in Real, the float centre being promoted when Real = double.
template <typename Key, typename Real>
__global__ __launch_bounds__(256) void morton_keys_kernel(const float4* __restrict__ spheres,
                                                          size_t n, Real minx, Real miny,
                                                          Real minz, Real sx, Real sy, Real sz,
                                                          Key* __restrict__ keys)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n;
         i += size_t(gridDim.x) * blockDim.x) {
        const float4 s = spheres[i];
        const Key x = static_cast<Key>(sx * (s.x - minx));
        const Key y = static_cast<Key>(sy * (s.y - miny));
        const Key z = static_cast<Key>(sz * (s.z - minz));
        keys[i] = Interleave<Key>::key(x, y, z);
    }
}

// Order-preserving float <-> uint map so that min/max can use integer atomics.
__device__ __forceinline__ uint32_t f2ord(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__host__ inline float ord2f(uint32_t u)
{
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

__global__ __launch_bounds__(256) void minmax_f4_kernel(const float4* __restrict__ v, size_t n,
                                                        uint32_t* __restrict__ out8)
{
    float lo[4] = { INFINITY, INFINITY, INFINITY, INFINITY };
    float hi[4] = { -INFINITY, -INFINITY, -INFINITY, -INFINITY };
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n;
         i += size_t(gridDim.x) * blockDim.x) {
        const float4 s = v[i];
        const float c[4] = { s.x, s.y, s.z, s.w };
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo[k] = fminf(lo[k], c[k]);
            hi[k] = fmaxf(hi[k], c[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    }
    __shared__ float s_lo[4][4], s_hi[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s_lo[wave][k] = lo[k];
            s_hi[wave][k] = hi[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        float l = s_lo[0][k], h = s_hi[0][k];
        for (int w = 1; w < 4; ++w) {
            l = fminf(l, s_lo[w][k]);
            h = fmaxf(h, s_hi[w][k]);
        }
        atomicMin(&out8[k], f2ord(l));
        atomicMax(&out8[4 + k], f2ord(h));
    }
}

grace_status minmax_f4(const float* d_v4, size_t n, float* h_mins4, float* h_maxs4,
                       hipStream_t stream)
{
    GRACE_REQUIRE(d_v4 && n > 0, "minmax: empty input");
    FrameGuard frame;
    GRACE_TRY(frame.begin(256, stream));
    uint32_t* d_out = Workspace::take<uint32_t>(8);
    GRACE_TRY_HIP(hipMemsetAsync(d_out, 0xFF, 16, stream));
    GRACE_TRY_HIP(hipMemsetAsync(d_out + 4, 0x00, 16, stream));
    minmax_f4_kernel<<<stream_grid(n, 256, 4), 256, 0, stream>>>(
        reinterpret_cast<const float4*>(d_v4), n, d_out);
    GRACE_CHECK_LAUNCH();
    uint32_t h[8];
    GRACE_TRY_HIP(hipMemcpyAsync(h, d_out, sizeof(h), hipMemcpyDeviceToHost, stream));
    GRACE_TRY_HIP(hipStreamSynchronize(stream));
    for (int k = 0; k < 4; ++k) {
        h_mins4[k] = ord2f(h[k]);
        h_maxs4[k] = ord2f(h[4 + k]);
    }
    return GRACE_OK;
}

template <typename Key, typename Real>
grace_status morton_keys(const float* d_spheres, size_t n, const Real* bot, const Real* top,
                         Key* d_keys, hipStream_t stream)
{
    GRACE_REQUIRE(d_spheres && d_keys && bot && top, "morton_keys: null pointer");
    if (n == 0) return GRACE_OK;
    // Host-side scale in Real3 precision, integer span (morton.cuh:104-113).
    const int span = sizeof(Key) > 4 ? (1u << 21) - 1 : (1u << 10) - 1;
    const Real sx = span / (top[0] - bot[0]);
    const Real sy = span / (top[1] - bot[1]);
    const Real sz = span / (top[2] - bot[2]);
    morton_keys_kernel<Key, Real><<<stream_grid(n, 256), 256, 0, stream>>>(
        reinterpret_cast<const float4*>(d_spheres), n, bot[0], bot[1], bot[2], sx, sy, sz,
        d_keys);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

// Generic points: n records of `stride` Elem each, x y z first (float3, float4, double3,
// double4).  CentroidSphere (generic/functors/centroid.h:33-40) narrows the co-ordinates to
// float BEFORE the key arithmetic, for double input too -- tests/morton_key_kernel/
// 63bit_keys.cu:52-58 spells out that the host must cast to match.
template <typename Key, typename Elem, typename Real>
__global__ __launch_bounds__(256) void morton_keys_points_kernel(const Elem* __restrict__ pts,
                                                                 size_t n, int stride, Real minx,
                                                                 Real miny, Real minz, Real sx,
                                                                 Real sy, Real sz,
                                                                 Key* __restrict__ keys)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n;
         i += size_t(gridDim.x) * blockDim.x) {
        const Elem* q = pts + i * size_t(stride);
        const float cx = static_cast<float>(q[0]), cy = static_cast<float>(q[1]),
                    cz = static_cast<float>(q[2]);
        const Key x = static_cast<Key>(sx * (cx - minx));
        const Key y = static_cast<Key>(sy * (cy - miny));
        const Key z = static_cast<Key>(sz * (cz - minz));
        keys[i] = Interleave<Key>::key(x, y, z);
    }
}

template <typename Key, typename Real>
grace_status morton_keys_points(const void* d_points, size_t n, int is_double, int stride,
                                const Real* bot, const Real* top, Key* d_keys,
                                hipStream_t stream)
{
    GRACE_REQUIRE(d_points && d_keys && bot && top, "morton_keys (points): null pointer");
    GRACE_REQUIRE(stride >= 3 && stride <= 16, "morton_keys (points): elements per point must be 3..16");
    if (n == 0) return GRACE_OK;
    const int span = sizeof(Key) > 4 ? (1u << 21) - 1 : (1u << 10) - 1;
    const Real sx = span / (top[0] - bot[0]);
    const Real sy = span / (top[1] - bot[1]);
    const Real sz = span / (top[2] - bot[2]);
    if (is_double)
        morton_keys_points_kernel<Key, double, Real><<<stream_grid(n, 256), 256, 0, stream>>>(
            static_cast<const double*>(d_points), n, stride, bot[0], bot[1], bot[2], sx, sy, sz,
            d_keys);
    else
        morton_keys_points_kernel<Key, float, Real><<<stream_grid(n, 256), 256, 0, stream>>>(
            static_cast<const float*>(d_points), n, stride, bot[0], bot[1], bot[2], sx, sy, sz,
            d_keys);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

// min / max of the float-narrowed x y z of generic points (compute_centroids + min_vec3 /
// max_vec3 with CentroidSphere on double4 input, kernels/morton.cuh:139-174).
template <typename Elem>
__global__ __launch_bounds__(256) void points_minmax_kernel(const Elem* __restrict__ pts, size_t n,
                                                            int stride, uint32_t* __restrict__ out8)
{
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n;
         i += size_t(gridDim.x) * blockDim.x) {
        const Elem* q = pts + i * size_t(stride);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float c = static_cast<float>(q[k]);
            lo[k] = fminf(lo[k], c);
            hi[k] = fmaxf(hi[k], c);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    }
    __shared__ float s_lo[4][3], s_hi[4][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { s_lo[wave][k] = lo[k]; s_hi[wave][k] = hi[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        float l = s_lo[0][k], h = s_hi[0][k];
        for (int w = 1; w < 4; ++w) { l = fminf(l, s_lo[w][k]); h = fmaxf(h, s_hi[w][k]); }
        atomicMin(&out8[k], f2ord(l));
        atomicMax(&out8[4 + k], f2ord(h));
    }
}

// ---- triangle primitives: {v, e1, e2}, 9 floats (tests/profile_trace_triangle/triangle.cuh:11-25)
// TriangleCentroid (triangle.cuh:92-102): v + (1./3.) * (e1 + e2), all fp32 (the scalar binds
// to operator*(float, float3), tests/helper/vector_math.cuh:34-37).
__device__ __forceinline__ void tri_centroid(const float* __restrict__ t, float* c)
{
    const float third = float(1. / 3.);
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] = t[k] + third * (t[3 + k] + t[6 + k]);
}

__global__ __launch_bounds__(256) void tri_minmax_kernel(const float* __restrict__ tris, size_t n,
                                                         uint32_t* __restrict__ out8)
{
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n;
         i += size_t(gridDim.x) * blockDim.x) {
        float c[3];
        tri_centroid(tris + 9 * i, c);
#pragma unroll
        for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], c[k]); hi[k] = fmaxf(hi[k], c[k]); }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    }
    __shared__ float s_lo[4][3], s_hi[4][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { s_lo[wave][k] = lo[k]; s_hi[wave][k] = hi[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        float l = s_lo[0][k], h = s_hi[0][k];
        for (int w = 1; w < 4; ++w) { l = fminf(l, s_lo[w][k]); h = fmaxf(h, s_hi[w][k]); }
        atomicMin(&out8[k], f2ord(l));
        atomicMax(&out8[4 + k], f2ord(h));
    }
}

__global__ __launch_bounds__(256) void morton_keys_tri_kernel(const float* __restrict__ tris, size_t n,
                                                              float minx, float miny, float minz,
                                                              float sx, float sy, float sz,
                                                              uint32_t* __restrict__ keys)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n;
         i += size_t(gridDim.x) * blockDim.x) {
        float c[3];
        tri_centroid(tris + 9 * i, c);
        const uint32_t x = static_cast<uint32_t>(sx * (c[0] - minx));
        const uint32_t y = static_cast<uint32_t>(sy * (c[1] - miny));
        const uint32_t z = static_cast<uint32_t>(sz * (c[2] - minz));
        keys[i] = Interleave<uint32_t>::key(x, y, z);
    }
}

} // namespace

extern "C" {

grace_status grace_centroid_bounds_tri(const float* d_tris, size_t n, float* h_bot, float* h_top,
                                       grace_stream stream)
{
    GRACE_REQUIRE(d_tris && n > 0 && h_bot && h_top, "centroid_bounds_tri: bad argument");
    hipStream_t st = as_stream(stream);
    FrameGuard frame;
    GRACE_TRY(frame.begin(256, st));
    uint32_t* d_out = Workspace::take<uint32_t>(8);
    GRACE_TRY_HIP(hipMemsetAsync(d_out, 0xFF, 16, st));
    GRACE_TRY_HIP(hipMemsetAsync(d_out + 4, 0x00, 16, st));
    tri_minmax_kernel<<<stream_grid(n, 256, 4), 256, 0, st>>>(d_tris, n, d_out);
    GRACE_CHECK_LAUNCH();
    uint32_t h[8];
    GRACE_TRY_HIP(hipMemcpyAsync(h, d_out, sizeof(h), hipMemcpyDeviceToHost, st));
    GRACE_TRY_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < 3; ++k) { h_bot[k] = ord2f(h[k]); h_top[k] = ord2f(h[4 + k]); }
    return GRACE_OK;
}

grace_status grace_morton_keys30_tri(const float* d_tris, size_t n, const float* h_bot,
                                     const float* h_top, uint32_t* d_keys, grace_stream stream)
{
    GRACE_REQUIRE(d_tris && d_keys && h_bot && h_top, "morton_keys_tri: null pointer");
    if (n == 0) return GRACE_OK;
    const int span = (1u << 10) - 1;
    const float sx = span / (h_top[0] - h_bot[0]);
    const float sy = span / (h_top[1] - h_bot[1]);
    const float sz = span / (h_top[2] - h_bot[2]);
    morton_keys_tri_kernel<<<stream_grid(n, 256), 256, 0, as_stream(stream)>>>(
        d_tris, n, h_bot[0], h_bot[1], h_bot[2], sx, sy, sz, d_keys);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

grace_status grace_minmax_f4(const float* d_v4, size_t n, float* h_mins4, float* h_maxs4,
                             grace_stream stream)
{
    GRACE_REQUIRE(h_mins4 && h_maxs4, "minmax: null output");
    return minmax_f4(d_v4, n, h_mins4, h_maxs4, as_stream(stream));
}

grace_status grace_centroid_bounds_f4(const float* d_spheres, size_t n, float* h_bot,
                                      float* h_top, grace_stream stream)
{
    GRACE_REQUIRE(h_bot && h_top, "centroid_bounds: null output");
    float lo[4], hi[4];
    GRACE_TRY(minmax_f4(d_spheres, n, lo, hi, as_stream(stream)));
    for (int k = 0; k < 3; ++k) {
        h_bot[k] = lo[k];
        h_top[k] = hi[k];
    }
    return GRACE_OK;
}

grace_status grace_morton_keys30_f4(const float* d_spheres, size_t n, const float* h_bot,
                                    const float* h_top, uint32_t* d_keys, grace_stream stream)
{
    return morton_keys<uint32_t, float>(d_spheres, n, h_bot, h_top, d_keys, as_stream(stream));
}

grace_status grace_morton_keys63_f4(const float* d_spheres, size_t n, const float* h_bot,
                                    const float* h_top, uint64_t* d_keys, grace_stream stream)
{
    return morton_keys<uint64_t, float>(d_spheres, n, h_bot, h_top, d_keys, as_stream(stream));
}

grace_status grace_morton_keys63_f4_d3(const float* d_spheres, size_t n, const double* h_bot,
                                       const double* h_top, uint64_t* d_keys,
                                       grace_stream stream)
{
    return morton_keys<uint64_t, double>(d_spheres, n, h_bot, h_top, d_keys, as_stream(stream));
}

grace_status grace_morton_keys30_f4_d3(const float* d_spheres, size_t n, const double* h_bot,
                                       const double* h_top, uint32_t* d_keys,
                                       grace_stream stream)
{
    return morton_keys<uint32_t, double>(d_spheres, n, h_bot, h_top, d_keys, as_stream(stream));
}

grace_status grace_centroid_bounds_points(const void* d_points, size_t n, int is_double,
                                          int elems_per_point, float* h_bot, float* h_top,
                                          grace_stream stream)
{
    GRACE_REQUIRE(d_points && n > 0 && h_bot && h_top, "centroid_bounds (points): bad argument");
    GRACE_REQUIRE(elems_per_point >= 3 && elems_per_point <= 16,
                  "centroid_bounds (points): elements per point must be 3..16");
    hipStream_t st = as_stream(stream);
    FrameGuard frame;
    GRACE_TRY(frame.begin(256, st));
    uint32_t* d_out = Workspace::take<uint32_t>(8);
    GRACE_TRY_HIP(hipMemsetAsync(d_out, 0xFF, 16, st));
    GRACE_TRY_HIP(hipMemsetAsync(d_out + 4, 0x00, 16, st));
    if (is_double)
        points_minmax_kernel<double><<<stream_grid(n, 256, 4), 256, 0, st>>>(
            static_cast<const double*>(d_points), n, elems_per_point, d_out);
    else
        points_minmax_kernel<float><<<stream_grid(n, 256, 4), 256, 0, st>>>(
            static_cast<const float*>(d_points), n, elems_per_point, d_out);
    GRACE_CHECK_LAUNCH();
    uint32_t h[8];
    GRACE_TRY_HIP(hipMemcpyAsync(h, d_out, sizeof(h), hipMemcpyDeviceToHost, st));
    GRACE_TRY_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < 3; ++k) { h_bot[k] = ord2f(h[k]); h_top[k] = ord2f(h[4 + k]); }
    return GRACE_OK;
}

grace_status grace_morton_keys30_points_d3(const void* d_points, size_t n, int is_double,
                                           int elems_per_point, const double* h_bot,
                                           const double* h_top, uint32_t* d_keys,
                                           grace_stream stream)
{
    return morton_keys_points<uint32_t, double>(d_points, n, is_double, elems_per_point, h_bot,
                                                h_top, d_keys, as_stream(stream));
}

grace_status grace_morton_keys63_points_d3(const void* d_points, size_t n, int is_double,
                                           int elems_per_point, const double* h_bot,
                                           const double* h_top, uint64_t* d_keys,
                                           grace_stream stream)
{
    return morton_keys_points<uint64_t, double>(d_points, n, is_double, elems_per_point, h_bot,
                                                h_top, d_keys, as_stream(stream));
}

grace_status grace_morton_keys30_points(const void* d_points, size_t n, int is_double,
                                        int elems_per_point, const float* h_bot,
                                        const float* h_top, uint32_t* d_keys, grace_stream stream)
{
    return morton_keys_points<uint32_t, float>(d_points, n, is_double, elems_per_point, h_bot, h_top,
                                               d_keys, as_stream(stream));
}

grace_status grace_morton_keys63_points(const void* d_points, size_t n, int is_double,
                                        int elems_per_point, const float* h_bot,
                                        const float* h_top, uint64_t* d_keys, grace_stream stream)
{
    return morton_keys_points<uint64_t, float>(d_points, n, is_double, elems_per_point, h_bot, h_top,
                                               d_keys, as_stream(stream));
}

} // extern "C"
