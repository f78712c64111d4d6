// Adjacent-primitive "delta" metrics that define the ALBVH hierarchy, for gfx950.
// Replaces compute_deltas_kernel (reference include/grace/cuda/kernels/albvh.cuh:33-47) with
// the functors of include/grace/generic/functors/albvh.h:17-126.  deltas[i] = delta(i - 1),
// i in [0, n]; both ends are sentinels.  HBM-bound streaming: each lane loads its own
// element (16 B float4 or 4/8 B key) and takes the right neighbour's from the next lane
// (wave shuffle), so the algorithmic 36 B/sphere falls to ~20 B/sphere of real traffic.
// No FMA contraction (-ffp-contract=off): the Euclidean delta decides the tree topology.
#include "common.hpp"

using namespace grace_hip;

namespace {

enum { DELTA_EUCLID = 0, DELTA_AREA = 1 };

template <int KIND>
__device__ __forceinline__ float sphere_delta(const float4 a, const float4 b)
{
    if (KIND == DELTA_EUCLID) {
        // generic/functors/albvh.h:77-79
        return (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y) + (a.z - b.z) * (a.z - b.z);
    } else {
        // generic/functors/albvh.h:107-118 with AABBSphere (generic/functors/aabb.h:9-26)
        const float Lx = fmaxf(a.x + a.w, b.x + b.w) - fminf(a.x - a.w, b.x - b.w);
        const float Ly = fmaxf(a.y + a.w, b.y + b.w) - fminf(a.y - a.w, b.y - b.w);
        const float Lz = fmaxf(a.z + a.w, b.z + b.w) - fminf(a.z - a.w, b.z - b.w);
        return (Lx * Ly) + (Lx * Lz) + (Ly * Lz);
    }
}

// Thread t produces deltas[t + 1] = delta(t) = metric(prim t, prim t + 1), t in [0, n-1);
// deltas[0] and deltas[n] are the sentinels.
template <int KIND>
__global__ __launch_bounds__(256) void sphere_deltas_kernel(const float4* __restrict__ s,
                                                            size_t n, float* __restrict__ deltas)
{
    const size_t t = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    const int lane = threadIdx.x & 63;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < n) a = s[t];
    float4 b;
    b.x = __shfl_down(a.x, 1);
    b.y = __shfl_down(a.y, 1);
    b.z = __shfl_down(a.z, 1);
    b.w = __shfl_down(a.w, 1);
    if (lane == 63 && t + 1 < n) b = s[t + 1];
    if (t == 0) deltas[0] = INFINITY;
    if (t < n) deltas[t + 1] = (t + 1 < n) ? sphere_delta<KIND>(a, b) : INFINITY;
}

// DeltaEuclidean / DeltaSurfaceArea on double4 spheres (generic/functors/albvh.h:44-74,
// 84-126).  Euclidean: differences and products in double, the sum narrowed to the float the
// functor returns.  Surface area: AABBSphere forms centre -+ radius in double and narrows the
// corners to float3 (generic/functors/aabb.h:9-26); the merged extents and the area are fp32.
// Out = float, or double for a device_vector<double> of deltas (build_tree<double4>,
// tests/helper/tree.cuh:20-24: the float value widened).  32 B per sphere, two reads.
template <int KIND, typename Out>
__global__ __launch_bounds__(256) void sphere_deltas_d4_kernel(const double* __restrict__ s, size_t n,
                                                               Out* __restrict__ deltas)
{
    const size_t t = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    if (t == 0) deltas[0] = Out(INFINITY);
    if (t >= n) return;
    if (t + 1 >= n) { deltas[t + 1] = Out(INFINITY); return; }
    const double* a = s + 4 * t;
    const double* b = a + 4;
    if (KIND == DELTA_EUCLID) {
        const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
        deltas[t + 1] = Out(float(dx * dx + dy * dy + dz * dz));
    } else {
        float L[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float bi = float(a[k] - a[3]), ti = float(a[k] + a[3]);
            const float bj = float(b[k] - b[3]), tj = float(b[k] + b[3]);
            L[k] = fmaxf(ti, tj) - fminf(bi, bj);
        }
        deltas[t + 1] = Out((L[0] * L[1]) + (L[0] * L[2]) + (L[1] * L[2]));
    }
}

template <typename Key>
__global__ __launch_bounds__(256) void xor_deltas_kernel(const Key* __restrict__ keys, size_t n,
                                                         Key* __restrict__ deltas)
{
    const size_t t = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    if (t == 0) deltas[0] = Key(~Key(0));
    if (t < n) deltas[t + 1] = (t + 1 < n) ? Key(keys[t] ^ keys[t + 1]) : Key(~Key(0));
}

} // namespace

extern "C" {

grace_status grace_deltas_euclid_f4(const float* d_spheres, size_t n, float* d_deltas,
                                    grace_stream stream)
{
    GRACE_REQUIRE(d_spheres && d_deltas && n > 0, "deltas: null pointer or empty input");
    sphere_deltas_kernel<DELTA_EUCLID><<<ceil_div(n, 256), 256, 0, as_stream(stream)>>>(
        reinterpret_cast<const float4*>(d_spheres), n, d_deltas);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

grace_status grace_deltas_area_f4(const float* d_spheres, size_t n, float* d_deltas,
                                  grace_stream stream)
{
    GRACE_REQUIRE(d_spheres && d_deltas && n > 0, "deltas: null pointer or empty input");
    sphere_deltas_kernel<DELTA_AREA><<<ceil_div(n, 256), 256, 0, as_stream(stream)>>>(
        reinterpret_cast<const float4*>(d_spheres), n, d_deltas);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

#define GRACE_DELTAS_D4(NAME, KIND, OUT)                                                        \
    grace_status NAME(const double* d_spheres, size_t n, OUT* d_deltas, grace_stream stream)     \
    {                                                                                            \
        GRACE_REQUIRE(d_spheres && d_deltas && n > 0, "deltas: null pointer or empty input");    \
        sphere_deltas_d4_kernel<KIND, OUT><<<ceil_div(n, 256), 256, 0, as_stream(stream)>>>(     \
            d_spheres, n, d_deltas);                                                             \
        GRACE_CHECK_LAUNCH();                                                                    \
        return GRACE_OK;                                                                         \
    }
GRACE_DELTAS_D4(grace_deltas_euclid_d4, DELTA_EUCLID, float)
GRACE_DELTAS_D4(grace_deltas_euclid_d4_f64, DELTA_EUCLID, double)
GRACE_DELTAS_D4(grace_deltas_area_d4, DELTA_AREA, float)
GRACE_DELTAS_D4(grace_deltas_area_d4_f64, DELTA_AREA, double)
#undef GRACE_DELTAS_D4

grace_status grace_deltas_xor_u32(const uint32_t* d_keys, size_t n, uint32_t* d_deltas,
                                  grace_stream stream)
{
    GRACE_REQUIRE(d_keys && d_deltas && n > 0, "deltas: null pointer or empty input");
    xor_deltas_kernel<uint32_t><<<ceil_div(n, 256), 256, 0, as_stream(stream)>>>(d_keys, n,
                                                                               d_deltas);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

grace_status grace_deltas_xor_u64(const uint64_t* d_keys, size_t n, uint64_t* d_deltas,
                                  grace_stream stream)
{
    GRACE_REQUIRE(d_keys && d_deltas && n > 0, "deltas: null pointer or empty input");
    xor_deltas_kernel<uint64_t><<<ceil_div(n, 256), 256, 0, as_stream(stream)>>>(d_keys, n,
                                                                               d_deltas);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

} // extern "C"
