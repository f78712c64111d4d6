// grace/hip/trace_core.hpp -- the GENERIC, functor-parameterised traversal for user-defined
// primitives and ray payloads, as a header-only HIP template (compile the including file with
// hipcc --offload-arch=gfx950).  It is the extension point the reference exposes as
// grace::trace / grace::trace_texref (include/grace/cuda/kernels/bintree_trace.cuh:214-367);
// user functors cannot cross the C ABI, so this one piece stays a template.  The built-in
// SPH / triangle instantiations behind libgrace_hip.so are faster (beam culling, treelet
// sweeps); this kernel is the plain packet walk with the reference's functor contract:
//
//   init(smem)                                   once per workgroup, before a barrier
//   ray_entry(ray_idx, ray, ray_data, smem)      ray_data is value-initialised first
//   intersect(ray, prim, ray_data, i, smem) -> bool      i = index inside the leaf
//   on_hit(ray_idx, ray, ray_data, prim_idx, prim, i, smem)
//   ray_exit(ray_idx, ray, ray_data, smem)
//
// Same semantics as the reference kernel (bintree_trace.cuh:52-197): one stack per packet,
// a child is pushed when ANY ray of the packet hits its box, right child first, and every
// ray of the packet is tested against every primitive of every leaf the packet enters, in
// ascending primitive order.  A packet is a 64-lane wavefront; N_rays need only be a
// multiple of 32 as in the reference (tail lanes are masked).
// This file holds the kernel, the launch over raw pointers and the reference's stock functors;
// it knows no container.  Include grace/hip/trace.hpp (HIP-free mirror's containers) or
// grace/cuda/kernels/bintree_trace.cuh (thrust::device_vector) for grace::trace itself.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <stdexcept>

#include <cstdio>
#include <cstdlib>

#include "grace/ray.h"

namespace grace {
namespace gpu {

// Bounds-carrying view of the user's LDS bytes (role of util/bound_iter.cuh:18-229).
template <typename T>
class BoundIter {
public:
    __device__ BoundIter(char* begin, size_t bytes)
        : p_(reinterpret_cast<T*>(begin)), end_(begin + bytes) {}
    template <typename U>
    __device__ BoundIter(const BoundIter<U>& o) : p_(reinterpret_cast<T*>(o.raw())), end_(o.raw_end()) {}
    __device__ T& operator[](ptrdiff_t i) const { return p_[i]; }
    __device__ T& operator*() const { return *p_; }
    __device__ BoundIter operator+(ptrdiff_t n) const { BoundIter r(*this); r.p_ += n; return r; }
    __device__ char* raw() const { return reinterpret_cast<char*>(p_); }
    __device__ char* raw_end() const { return end_; }
    __device__ size_t size() const { return size_t(end_ - reinterpret_cast<char*>(p_)) / sizeof(T); }
private:
    T* p_;
    char* end_;
};

__device__ __forceinline__ int imin_(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax_(int a, int b) { return a > b ? a : b; }

// Two-child slab test, include/grace/cuda/device/intersect.cuh:10-40 (integer min/max on the
// float bit patterns, device/intrinsics.cuh:8-51).  bit0 = right, bit1 = left.
__device__ __forceinline__ int AABBs_hit(const float ix, const float iy, const float iz,
                                         const float ox, const float oy, const float oz,
                                         const float len, const ::float4 L, const ::float4 R,
                                         const ::float4 Z)
{
    const float bxL = (L.x - ox) * ix, txL = (L.y - ox) * ix, byL = (L.z - oy) * iy, tyL = (L.w - oy) * iy;
    const float bzL = (Z.x - oz) * iz, tzL = (Z.y - oz) * iz;
    const float bxR = (R.x - ox) * ix, txR = (R.y - ox) * ix, byR = (R.z - oy) * iy, tyR = (R.w - oy) * iy;
    const float bzR = (Z.z - oz) * iz, tzR = (Z.w - oz) * iz;
    const int zero = __float_as_int(0.0f), il = __float_as_int(len);
    const int tminL = imax_(imax_(__float_as_int(fminf(bxL, txL)), __float_as_int(fminf(byL, tyL))),
                            imax_(imin_(__float_as_int(bzL), __float_as_int(tzL)), zero));
    const int tmaxL = imin_(imin_(__float_as_int(fmaxf(bxL, txL)), __float_as_int(fmaxf(byL, tyL))),
                            imin_(imax_(__float_as_int(bzL), __float_as_int(tzL)), il));
    const int tminR = imax_(imax_(__float_as_int(fminf(bxR, txR)), __float_as_int(fminf(byR, tyR))),
                            imax_(imin_(__float_as_int(bzR), __float_as_int(tzR)), zero));
    const int tmaxR = imin_(imin_(__float_as_int(fmaxf(bxR, txR)), __float_as_int(fmaxf(byR, tyR))),
                            imin_(imax_(__float_as_int(bzR), __float_as_int(tzR)), il));
    return int(__int_as_float(tmaxR) >= __int_as_float(tminR))
         + 2 * int(__int_as_float(tmaxL) >= __int_as_float(tminL));
}

template <typename RayData, typename TPrimitive, typename Init, typename Intersection,
          typename OnHit, typename OnRayEntry, typename OnRayExit>
__global__ __launch_bounds__(256) void trace_kernel(const Ray* __restrict__ rays, const int n_rays,
                                                    const ::float4* __restrict__ nodes,
                                                    const int n_nodes,
                                                    const ::int4* __restrict__ leaves,
                                                    const int* __restrict__ root_index,
                                                    const TPrimitive* __restrict__ primitives,
                                                    const size_t user_smem_bytes, Init init,
                                                    Intersection intersect, OnHit on_hit,
                                                    OnRayEntry ray_entry, OnRayExit ray_exit,
                                                    int* __restrict__ status)
{
    extern __shared__ __align__(16) char smem_trace[];
    const BoundIter<char> sm_iter_usr(smem_trace, user_smem_bytes);
    init(sm_iter_usr);
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int packet = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (packet * 64 >= n_rays) return;
    const int slot = packet * 64 + lane;
    const bool valid = slot < n_rays;
    const int ray_index = valid ? slot : n_rays - 1;
    const Ray ray = rays[ray_index];
    RayData ray_data = {};
    if (valid) ray_entry(ray_index, ray, ray_data, sm_iter_usr);
    const float ix = 1.f / ray.dx, iy = 1.f / ray.dy, iz = 1.f / ray.dz;

    int stk0 = 0, stk1 = 0, sp = -1;     // packet stack: entry e in lane e & 63 of stk0 / stk1
    bool overflow = false;
    auto push = [&](const int v) {
        if (sp >= 127) { overflow = true; return; }
        ++sp;
        if (sp < 64) stk0 = (lane == sp) ? v : stk0; else stk1 = (lane == sp - 64) ? v : stk1;
    };
    push(*root_index);
    while (sp >= 0) {
        const int idx = sp < 64 ? __builtin_amdgcn_readlane(stk0, sp)
                                : __builtin_amdgcn_readlane(stk1, sp - 64);
        --sp;
        if (idx < n_nodes) {
            const ::float4* np = nodes + 4 * size_t(idx);
            const ::float4 n0 = np[0], L = np[1], R = np[2], Z = np[3];
            const int lr = AABBs_hit(ix, iy, iz, ray.ox, ray.oy, ray.oz, ray.length, L, R, Z);
            if (__builtin_amdgcn_ballot_w64(lr & 1)) push(__float_as_int(n0.y));
            if (__builtin_amdgcn_ballot_w64(lr >= 2)) push(__float_as_int(n0.x));
        } else {
            const ::int4 leaf = leaves[idx - n_nodes];
            for (int i = 0; i < leaf.y; ++i) {
                const TPrimitive prim = primitives[leaf.x + i];
                if (valid && intersect(ray, prim, ray_data, i, sm_iter_usr))
                    on_hit(ray_index, ray, ray_data, leaf.x + i, prim, i, sm_iter_usr);
            }
        }
    }
    if (overflow && lane == 0) *status = 1;
    if (valid) ray_exit(ray_index, ray, ray_data, sm_iter_usr);
}

} // namespace gpu

// Launch of the generic kernel over raw device pointers; the public grace::trace /
// grace::trace_texref overloads (grace/hip/trace.hpp over grace::device_vector,
// grace/cuda/kernels/bintree_trace.cuh over thrust::device_vector) forward here.
namespace detail {
template <typename RayData, typename TPrimitive, typename Init, typename Intersection,
          typename OnHit, typename OnRayEntry, typename OnRayExit>
inline void trace_launch(const Ray* d_rays, const size_t N_rays, const TPrimitive* d_prims,
                         const void* d_nodes, const size_t n_nodes, const void* d_leaves,
                         const int* d_root_index, const size_t user_smem_bytes, Init init,
                         Intersection intersect, OnHit on_hit, OnRayEntry ray_entry,
                         OnRayExit ray_exit)
{
    // include/grace/cuda/kernels/bintree_trace.cuh:231-238
    if (N_rays % 32 != 0)
        throw std::invalid_argument("Number of rays must be a multiple of the warp size (32).");
    int* d_status = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d_status), sizeof(int)) != hipSuccess
        || hipMemset(d_status, 0, sizeof(int)) != hipSuccess)
        throw std::runtime_error("grace::trace: hipMalloc failed");
    const int n_packets = int((N_rays + 63) / 64);
    gpu::trace_kernel<RayData><<<(n_packets + 3) / 4, 256, user_smem_bytes, 0>>>(
        d_rays, int(N_rays), static_cast<const ::float4*>(d_nodes), int(n_nodes),
        static_cast<const ::int4*>(d_leaves), d_root_index, d_prims, user_smem_bytes, init,
        intersect, on_hit, ray_entry, ray_exit, d_status);
    int h_status = 0;
    const hipError_t e = hipMemcpy(&h_status, d_status, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(d_status);
    if (e != hipSuccess) { // include/grace/error.h:40-56
        std::fprintf(stderr, "**** GRACE HIP Error ****\n%s\n", hipGetErrorString(e));
        std::exit(int(e));
    }
    if (h_status) throw std::runtime_error("grace::trace: packet stack (128 entries) exhausted");
}
} // namespace detail

// ---- the reference's stock functors and payloads, for composing custom traces -----------

template <typename T>
struct RayData_datum { T data; };                       // include/grace/generic/raydata.h:5-9
template <typename T, typename Real>
struct RayData_sphere { T data; Real b2, dist; };       // generic/raydata.h:11-16

struct Init_null {                                      // functors/trace.cuh:18-25
    __device__ void operator()(const gpu::BoundIter<char>) const {}
};
struct RayEntry_null {                                  // functors/trace.cuh:27-38
    template <typename RayData>
    __device__ void operator()(int, const Ray&, const RayData&, const gpu::BoundIter<char>) const {}
};
typedef RayEntry_null RayExit_null;

template <typename T>
struct RayEntry_from_array {                            // functors/trace.cuh:44-60
    const T* inits;
    explicit RayEntry_from_array(const T* p) : inits(p) {}
    template <typename RayData>
    __device__ void operator()(int ray_idx, const Ray&, RayData& rd, const gpu::BoundIter<char>) const
    { rd.data = inits[ray_idx]; }
};
template <typename T>
struct RayExit_to_array {                               // functors/trace.cuh:65-81
    T* store;
    explicit RayExit_to_array(T* p) : store(p) {}
    template <typename RayData>
    __device__ void operator()(int ray_idx, const Ray&, const RayData& rd, const gpu::BoundIter<char>) const
    { store[ray_idx] = rd.data; }
};
template <typename T>
struct InitGlobalToSmem {                               // functors/trace.cuh:87-112
    const T* src; int count;
    InitGlobalToSmem(const T* p, int n) : src(p), count(n) {}
    __device__ void operator()(const gpu::BoundIter<char> smem) const
    {
        gpu::BoundIter<T> dst = smem;
        for (int i = threadIdx.x; i < count; i += blockDim.x) dst[i] = src[i];
    }
};

// include/grace/generic/intersect.h:10-55 (Real = float)
__device__ __forceinline__ bool sphere_hit(const Ray& ray, const ::float4& s, float& b2, float& dot_p)
{
    const float px = s.x - ray.ox, py = s.y - ray.oy, pz = s.z - ray.oz;
    dot_p = px * ray.dx + py * ray.dy + pz * ray.dz;
    const float bx = px - dot_p * ray.dx, by = py - dot_p * ray.dy, bz = pz - dot_p * ray.dz;
    b2 = bx * bx + by * by + bz * bz;
    if (b2 >= s.w * s.w) return false;
    if (dot_p < 0.0f) return false;
    if (dot_p >= ray.length) return false;
    return true;
}

struct Intersect_sphere_bool {                          // functors/trace.cuh:118-131
    template <typename RayData>
    __device__ bool operator()(const Ray& ray, const ::float4& s, const RayData&, int,
                               const gpu::BoundIter<char>) const
    { float b2, d; return sphere_hit(ray, s, b2, d); }
};
struct Intersect_sphere_b2dist {                        // functors/trace.cuh:134-145
    template <typename RayData>
    __device__ bool operator()(const Ray& ray, const ::float4& s, RayData& rd, int,
                               const gpu::BoundIter<char>) const
    { return sphere_hit(ray, s, rd.b2, rd.dist); }
};
struct OnHit_increment {                                // functors/trace.cuh:150-161
    template <typename RayData, typename TPrim>
    __device__ void operator()(int, const Ray&, RayData& rd, int, const TPrim&, int,
                               const gpu::BoundIter<char>) const
    { ++rd.data; }
};

// include/grace/generic/interpolate.h:11-39, device branch (fma in the table's precision)
template <typename TableIter>
__device__ __forceinline__ float lerp(float x, TableIter table, int N_table)
{
    int x_idx = static_cast<int>(x);
    if (x_idx >= N_table - 1) { x = static_cast<float>(N_table - 1); x_idx = N_table - 2; }
    const double y0 = table[x_idx], y1 = table[x_idx + 1];
    const double t = static_cast<double>(x) - x_idx;
    return static_cast<float>(__builtin_fma(t, y1 - y0, y0));
}

struct OnHit_sphere_cumulate {                          // functors/trace.cuh:163-193
    int N_table;
    explicit OnHit_sphere_cumulate(int n) : N_table(n) {}
    template <typename RayData>
    __device__ void operator()(int, const Ray&, RayData& rd, int, const ::float4& s, int,
                               const gpu::BoundIter<char> smem) const
    {
        gpu::BoundIter<double> Wk = smem;
        const float ir = 1.f / s.w;
        const float b = (N_table - 1) * (__builtin_sqrtf(rd.b2) * ir);
        float integral = lerp(b, Wk, N_table);
        integral *= (ir * ir);
        rd.data += integral;
    }
};

// functors/trace.cuh:196-235: the per-hit outputs of trace_sph (pass 2): the ray's running output
// position is its RayData.data (initialised by RayEntry_from_array with the ray's offset).
template <typename IndexType, typename Real>
struct OnHit_sphere_individual {
    IndexType* indices; Real* integrals; Real* distances; int N_table;
    OnHit_sphere_individual(IndexType* i, Real* w, Real* d, int n)
        : indices(i), integrals(w), distances(d), N_table(n) {}
    template <typename RayData>
    __device__ void operator()(int, const Ray&, RayData& rd, int sphere_idx, const ::float4& s, int,
                               const gpu::BoundIter<char> smem) const
    {
        gpu::BoundIter<double> Wk = smem;
        const float ir = 1.f / s.w;
        const float b = (N_table - 1) * (__builtin_sqrtf(rd.b2) * ir);
        Real integral = lerp(b, Wk, N_table);
        integral *= (ir * ir);
        indices[rd.data] = sphere_idx;
        integrals[rd.data] = integral;
        distances[rd.data] = rd.dist;
        ++rd.data;
    }
};

} // namespace grace
